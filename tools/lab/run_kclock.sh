# shader clock (cycle stamps / wall time) of the persistent GEMM with and without its epilogue / DMA, both tile geometries
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/lab_kclock.jsonl
: > $OUT
for r in 1 2; do for sh in siglip_fc1 mm_kv sq8k; do
  LAB_SHAPE=$sh timeout 300 tools/lab/gemm_lab kclock 10 | grep '^{' | sed "s/^{/{\"round\": $r, /" >> $OUT
done; done
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/lab_kclock.jsonl")]
tf = {(d["round"], d["shape"], d["variant"]): d["tflops"] for d in rows if "tflops" in d}
for d in rows:
    if "eff_clock_GHz" in d:
        t = tf[(d["round"], d["shape"], d["variant"])]
        peak = 256 * 4 * 1024 * d["eff_clock_GHz"] / 1e3
        print(d["round"], d["shape"], d["variant"], "TFLOP/s", round(t), "clock", d["eff_clock_GHz"], "peak at that clock", round(peak), "MFMA-busy", round(t / peak, 3))
PY
