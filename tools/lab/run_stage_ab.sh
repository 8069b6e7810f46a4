# same-box A/B of the lock-step epilogue staging (-DVIDI_W4_STAGE_LOCKSTEP=1 / 0): wall time, checksums, and the epilogue's cycle stamps
cd $GRAFT_REPO_ROOT
: > gpurun_out/lab_stage.jsonl
for r in 1 2 3; do for v in sl1 sl0; do
  for sh in siglip_o siglip_fc1 siglip_fc2; do
    for var in w4p w4p_brs w4p_bt; do LAB_SHAPE=$sh tools/lab/gemm_lab_$v $var 5 | sed "s/^{/{\"build\": \"$v\", /" >> gpurun_out/lab_stage.jsonl; done
  done
  LAB_SHAPE=mm_down tools/lab/gemm_lab_$v w4p 3 | sed "s/^{/{\"build\": \"$v\", /" >> gpurun_out/lab_stage.jsonl
done; done
for v in sl1 sl0; do LAB_SHAPE=siglip_o tools/lab/gemm_lab_$v stamps_epi 3 | sed "s/^{/{\"build\": \"$v\", /" >> gpurun_out/lab_stage.jsonl; done
python - <<'PY'
import json, collections
acc=collections.defaultdict(list); cs=collections.defaultdict(set)
for l in open("gpurun_out/lab_stage.jsonl"):
    if l.startswith("{"):
        d=json.loads(l)
        if "cycles_3" in d: print(d["build"], d["shape"], d["variant"], "epilogue cycles per tile %.0f" % (d["cycles_3"]/(d["tiles"]/256))); continue
        if "stamps" in d["variant"]: continue
        acc[(d["shape"],d["variant"],d["build"])].append(round(d["tflops"])); cs[(d["shape"],d["variant"])].add(d["checksum"])
for k in sorted(acc): print(k, acc[k])
print({k: len(v) for k,v in cs.items()}, "(1 = both builds give the same checksum)")
PY
