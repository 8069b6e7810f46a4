#!/bin/bash
# round-5 session 3c: gemvm with lane-masked X loads; 32-feature groups (VIDI_GEMVM_FG2); ring depths; then the configs[4] bench
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; export PYTHONPATH=$REPO
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemv_mfma" > $OUT/s3c_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/s3c_pytest.log
VIDI_GEMVM_FG2=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemv_mfma" > $OUT/s3c_pytest_fg2.log 2>&1; echo "pytest fg2 rc=$?"; tail -3 $OUT/s3c_pytest_fg2.log
: > $OUT/s3c_gemvm_variants.jsonl
for cfg in "libvidi_hip.so 0" "libvidi_hip.so 1" "libvidi_hip_gemvm_d12.so 0" "libvidi_hip_gemvm_d12.so 1" "libvidi_hip.so 0" "libvidi_hip.so 1"; do
  set -- $cfg
  VIDI_HIP_LIB=$REPO/vidi_amd/$1 VIDI_GEMVM_FG2=$2 timeout 120 python tools/bench_gemv_mfma.py 8 30 mfma | sed "s/^{/{\"lib\": \"$1\", \"fg2\": $2, /" >> $OUT/s3c_gemvm_variants.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/s3c_gemvm_variants.jsonl"):
    d = json.loads(l); print(d["lib"], "fg2", d["fg2"], d["shape"], d["M"], d["mfma_us"], d["mfma_TB/s"])
PY
CFG4="--fps 2 --queries 8 --ragged-prompts 24 52 --decode-steps 128 --steps 1 --warmup 1 --no-verify --no-cpu-baseline --no-preproc --no-kernel-timer"
for fg in 0 1; do
VIDI_GEMVM_FG2=$fg timeout 600 python bench.py $CFG4 > $OUT/s3c_cfg4_fg$fg.json 2> $OUT/s3c_cfg4.err; echo "cfg4 fg2=$fg rc=$?"
python tools/show_bench.py $OUT/s3c_cfg4_fg$fg.json 2>/dev/null | grep -E "value|stages" | head -4
done
