# 288 x 224 tile geometry (gemm_w4n.h) against the 256-wide persistent kernel on SigLIP's out_proj / fc2 shapes: checksums of Y (must be
# identical), TFLOP/s alternating over three rounds, cycle stamps; then the product's parity tests of the statistics path
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/lab_w4n.jsonl
: > $OUT
for r in 1 2 3; do for sh in siglip_o siglip_fc2; do
  LAB_SHAPE=$sh timeout 300 tools/lab/gemm_lab w4n 5 | grep '^{' | sed "s/^{/{\"round\": $r, /" >> $OUT
done; done
python - <<'PY'
import json, collections
acc = collections.defaultdict(list); ver = collections.defaultdict(set)
for l in open("gpurun_out/lab_w4n.jsonl"):
    d = json.loads(l)
    if "tflops" in d:
        acc[(d["shape"], d["variant"])].append(round(d["tflops"])); ver[(d["shape"], d["variant"])].add(d["vs_ref"])
    elif d["round"] == 1:
        tiles = d["tiles"] if "w4n" not in d["variant"] else 4 * -(-262440 // 224)
        print("stamps", d["shape"], d["variant"], "K loop / tile", round(d["cycles_1"] * d["blocks"] / tiles), "epilogue / tile", round(d["cycles_3"] * d["blocks"] / tiles), "clock", d.get("eff_clock_GHz"))
for k in sorted(acc): print(k, acc[k], sorted(ver[k]))
PY
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm_res_stats or gemm_ln or gemm_basic or test_gemm" 2>&1 | tail -5
