#!/bin/bash
# round-5 session 2: many-row cross-attention (parity + same-box A/B against the per-tile kernel), decode-batch projection launch-shape sweep
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; export PYTHONPATH=$REPO
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "attn_cross or gemv_mfma" > $OUT/s2_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/s2_pytest.log
: > $OUT/s2_xattn.jsonl
for rows in 1 0; do
  if [ $rows = 1 ]; then Z39="16 32 64"; Z304="3 6 12"; else Z39="10"; Z304="1"; fi
  VIDI_XATTN_ROWS=$rows timeout 300 python tools/bench_xattn.py --keys 90000 --lq 39 --iters 20 --zsplit $Z39 >> $OUT/s2_xattn.jsonl 2>> $OUT/s2_xattn.err
  VIDI_XATTN_ROWS=$rows timeout 300 python tools/bench_xattn.py --keys 90000 --lq 304 --iters 10 --zsplit $Z304 >> $OUT/s2_xattn.jsonl 2>> $OUT/s2_xattn.err
done
cat $OUT/s2_xattn.jsonl
: > $OUT/s2_gemvm_sweep.jsonl
for ks in 0 1 2 4 8; do for res in 1024 2048 4096; do
  VIDI_GEMVM_KS=$ks VIDI_GEMVM_RESIDENT_WAVES=$res timeout 120 python tools/bench_gemv_mfma.py 8 30 mfma >> $OUT/s2_gemvm_sweep.jsonl 2>> $OUT/s2_gemvm.err
done; done
python - <<'PY'
import json, collections
best = collections.defaultdict(list)
for l in open("gpurun_out/s2_gemvm_sweep.jsonl"):
    d = json.loads(l)
    best[(d["shape"], d["M"])].append((d["mfma_us"], d["env"].get("VIDI_GEMVM_KS"), d["env"].get("VIDI_GEMVM_RESIDENT_WAVES")))
for k, v in best.items():
    print(k, sorted(v)[:4], "worst", sorted(v)[-1])
PY
timeout 120 python tools/bench_gemv_mfma.py 8 30 > $OUT/s2_gemv_default.jsonl 2>> $OUT/s2_gemvm.err; cat $OUT/s2_gemv_default.jsonl
CFG4="--fps 2 --queries 8 --ragged-prompts 24 52 --decode-steps 128 --steps 1 --warmup 1 --no-verify --no-cpu-baseline --no-preproc --no-kernel-timer"
timeout 600 python bench.py $CFG4 > $OUT/s2_cfg4.json 2> $OUT/s2_cfg4.err; echo "cfg4 rc=$?"
python tools/show_bench.py $OUT/s2_cfg4.json 2>/dev/null | grep -E "value|stages" | head -4
