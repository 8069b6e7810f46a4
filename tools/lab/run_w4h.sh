# 256 x 128 tile geometry probe (tools/lab/gemm_w4h.h: half the accumulator file) against the 256 x 256 persistent kernel: K-loop-only
# rates (noepi), with the bias + residual epilogue, no-DMA ceiling, cycle stamps; Y (and the partial sums) must be bit-identical
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/lab_w4h.jsonl
: > $OUT
for r in 1 2 3; do for sh in siglip_o siglip_fc1 mm_kv; do
  LAB_SHAPE=$sh timeout 300 tools/lab/gemm_lab w4h 5 | grep '^{' | sed "s/^{/{\"round\": $r, /" >> $OUT
done; done
python - <<'PY'
import json, collections
acc = collections.defaultdict(list); ver = collections.defaultdict(set)
for l in open("gpurun_out/lab_w4h.jsonl"):
    d = json.loads(l)
    if "tflops" in d:
        acc[(d["shape"], d["variant"])].append(round(d["tflops"])); ver[(d["shape"], d["variant"])].add(d["vs_ref"])
    elif d["round"] == 1:
        print("stamps", d)
for k in sorted(acc): print(k, acc[k], sorted(ver[k]))
PY
