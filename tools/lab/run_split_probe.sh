cd $GRAFT_REPO_ROOT
: > gpurun_out/lab_split.jsonl
for sh in siglip_o o_n1024 o_rem128 siglip_fc2 fc2_n1024 fc2_rem128; do
  LAB_SHAPE=$sh tools/lab/gemm_lab_split split 5 >> gpurun_out/lab_split.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/lab_split.jsonl"):
    if l.startswith("{"):
        d=json.loads(l); print(d["shape"], d["variant"], "ms", round(d["ms"],4), "TF", round(d["tflops"]), d["vs_ref"])
PY
