# Round 6: where the tower GEMMs (K = 1 152) stand against their own K loop, on the CURRENT kernel, and the fabric-traffic claim re-measured.
#   part 1  whole kernel vs the same kernel without its epilogue (LabNoEpi: the K loop + tile switch only), per tower shape, product epilogue forms
#   part 2  tile order x shader clock x FETCH_SIZE on the gate|up shape (tools/lab/run_clock.sh): does 10x the fabric traffic cost clock?
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/lab_tower_bound.jsonl
: > $OUT
for r in 1 2 3; do
  for v in w4p_lnf1_bt w4p_lnf1_noepi; do for sh in siglip_fc1 siglip_qkv whisper_fc1; do
    LAB_SHAPE=$sh timeout 300 tools/lab/gemm_lab $v 5 | grep '^{' | sed "s/^{/{\"round\": $r, /" >> $OUT
  done; done
  for v in w4n_brs w4n_br_noepi w4p_brs_o1 w4p_br_noepi_o1; do for sh in siglip_o siglip_fc2; do
    LAB_SHAPE=$sh timeout 300 tools/lab/gemm_lab $v 5 | grep '^{' | sed "s/^{/{\"round\": $r, /" >> $OUT
  done; done
  for v in w4p w4p_noepi; do for sh in mm_kv mm_down mm_o; do
    LAB_SHAPE=$sh timeout 300 tools/lab/gemm_lab $v 5 | grep '^{' | sed "s/^{/{\"round\": $r, /" >> $OUT
  done; done
done
python - <<'PY'
import json, collections
acc = collections.defaultdict(list)
for l in open("gpurun_out/lab_tower_bound.jsonl"):
    d = json.loads(l)
    if "tflops" in d: acc[(d["shape"], d["variant"])].append(d["tflops"])
for k in sorted(acc): print(k, [round(x) for x in acc[k]])
PY
bash tools/lab/run_clock.sh 2>&1 | tail -40
