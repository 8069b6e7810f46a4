#!/bin/bash
# round-6 final evidence on ONE box: PMC traffic (keyed to this build's GEMM digest), driver-shaped bench, rocprofv3 kernel stats, MFMA-busy,
# the whole GPU suite under the guard-zone allocator, both dist modes with two ranks on the one GPU, configs[4] kernel stats, the other presets
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_round.sh pmc bench20 prof mfma canary dist2g
CFG4="--fps 2 --queries 8 --ragged-prompts 24 52 --decode-steps 128 --steps 1 --warmup 1 --no-verify --no-cpu-baseline --no-preproc --no-other-configs --no-kernel-timer"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cfg4 -o d8 -- python $REPO/bench.py $CFG4 > $OUT/prof_cfg4_bench.json 2> $OUT/prof_cfg4.err); echo "prof cfg4 rc=$?"
find $OUT/prof_cfg4 -name '*kernel_trace.csv' -delete; find $OUT/prof_cfg4 -name '*.db' -delete
for v in "--preset vidi_7b" "--decode-graph"; do
  n=$(echo $v | tr -d ' -' | cut -c1-24); timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-preproc --no-other-configs $v > $OUT/bench_$n.json 2> $OUT/bench_$n.err; echo "bench $v rc=$?"
  python tools/show_bench.py $OUT/bench_$n.json 2>/dev/null | grep -E "value|stages"
done
