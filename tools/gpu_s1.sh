#!/bin/bash
# round-5 session 1: new-kernel parity, decode-batch micro A/B, in-loop LayerNorm statistics lab, configs[4] decode A/B + kernel stats
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemv_mfma or gemm_ln_rows or ln_finalize" > $OUT/s1_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/s1_pytest.log
timeout 300 python tools/bench_gemv_mfma.py 8 40 > $OUT/s1_gemv_mfma.jsonl 2> $OUT/s1_gemv_mfma.err; echo "gemv bench rc=$?"; cat $OUT/s1_gemv_mfma.jsonl
bash tools/lab/run_lnstats.sh 2>&1 | tail -40
CFG4="--fps 2 --queries 8 --ragged-prompts 24 52 --decode-steps 128 --steps 1 --warmup 1 --no-verify --no-cpu-baseline --no-preproc"
for sw in 0 5; do
  VIDI_GEMV_MFMA_MIN_ROWS=$sw timeout 600 python bench.py $CFG4 > $OUT/s1_cfg4_mfma$sw.json 2> $OUT/s1_cfg4_mfma$sw.err; echo "cfg4 mfma_rows=$sw rc=$?"
  python tools/show_bench.py $OUT/s1_cfg4_mfma$sw.json 2>/dev/null | grep -E "value|stages|decode|query" | head -8
done
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s1_prof -o d8 -- python $REPO/bench.py $CFG4 --no-kernel-timer > $OUT/s1_prof_bench.json 2> $OUT/s1_prof.err); echo "prof rc=$?"
find $OUT/s1_prof -name '*kernel_trace.csv' -delete; find $OUT/s1_prof -name '*.db' -delete
find $OUT/s1_prof -name '*kernel_stats.csv' | head -2
