#!/bin/bash
# round-5 session 4: software-pipelined many-row cross-attention: parity, same-box A/B against the unpipelined build, end-to-end legs
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; export PYTHONPATH=$REPO
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "attn_cross" > $OUT/s4_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/s4_pytest.log
: > $OUT/s4_xattn.jsonl
for lib in libvidi_hip.so libvidi_hip_xrows_nopipe.so libvidi_hip.so libvidi_hip_xrows_nopipe.so; do
  VIDI_HIP_LIB=$REPO/vidi_amd/$lib timeout 300 python tools/bench_xattn.py --keys 90000 --lq 39 --iters 20 --zsplit 32 | sed "s/^{/{\"lib\": \"$lib\", /" >> $OUT/s4_xattn.jsonl
  VIDI_HIP_LIB=$REPO/vidi_amd/$lib timeout 300 python tools/bench_xattn.py --keys 90000 --lq 304 --iters 10 --zsplit 6 | sed "s/^{/{\"lib\": \"$lib\", /" >> $OUT/s4_xattn.jsonl
  VIDI_HIP_LIB=$REPO/vidi_amd/$lib timeout 300 python tools/bench_xattn.py --keys 90000 --lq 416 --iters 10 --zsplit 4 | sed "s/^{/{\"lib\": \"$lib\", /" >> $OUT/s4_xattn.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/s4_xattn.jsonl"):
    d = json.loads(l); print(d["lib"], "Lq", d["Lq"], "zsplit", d["zsplit"], round(d["ms"], 4), "ms", round(d["GBps"]), "GB/s", round(d["TFLOPs"]), "TFLOP/s")
PY
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_reference_golden.py tests/test_gpu_baseline_scale.py -q -x > $OUT/s4_pytest_model.log 2>&1; echo "pytest model rc=$?"; tail -3 $OUT/s4_pytest_model.log
