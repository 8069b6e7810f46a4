# same-box A/B of the epilogue form 2 (tools/lab/gemm_lab_epi2: -DVIDI_W4_EPI2=1, buffer-addressed stores / residual loads / partial-sum
# stores, interleaved row reductions) against the shipped form (tools/lab/gemm_lab), alternating builds; checksums must be identical
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/lab_epi2.jsonl
: > $OUT
run() { # build variant shape
  LAB_SHAPE=$3 timeout 120 tools/lab/$1 $2 5 | grep '^{' | sed "s/^{/{\"build\": \"$1\", /" >> $OUT
}
for r in 1 2 3; do
  for b in gemm_lab gemm_lab_epi2; do
    for sh in siglip_o siglip_fc2; do run $b w4p_brs $sh; run $b w4p_br $sh; done
    for sh in siglip_fc1 whisper_fc1; do run $b w4p_bt $sh; run $b w4p_be $sh; done
    for sh in siglip_qkv mm_o mm_down mm_kv; do run $b w4p $sh; done
    run $b w4p_geglu mm_gateup
  done
done
for b in gemm_lab gemm_lab_epi2; do for sh in siglip_o siglip_fc1; do run $b w4p_brs_stamps $sh; run $b w4p_bt_stamps $sh; run $b w4p_stamps $sh; done; done
python - <<'PY'
import json, collections
acc=collections.defaultdict(list); cs=collections.defaultdict(set)
for l in open("gpurun_out/lab_epi2.jsonl"):
    if l.startswith("{"):
        d=json.loads(l)
        if "tflops" in d:
            acc[(d["shape"],d["variant"],d["build"])].append(round(d["tflops"])); cs[(d["shape"],d["variant"])].add(d["checksum"])
        else:
            print("stamps", d["build"], d["shape"], d["variant"], "kloop/tile", round(d["cycles_1"]*256/d["tiles"]) if d["blocks"]==256 else None, "epi/tile", round(d["cycles_3"]*d["blocks"]/d["tiles"]), "clock", d.get("eff_clock_GHz"))
for k in sorted(acc): print(k, acc[k])
print({str(k): len(v) for k,v in cs.items()}, "(1 = both builds give the same checksum)")
PY
