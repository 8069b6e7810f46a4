# same-box A/B of the lock-step (packed) GELU epilogues: two lab builds (-DVIDI_GELU_LOCKSTEP=1 / 0), alternating, checksums compared
cd $GRAFT_REPO_ROOT
: > gpurun_out/lab_gelu.jsonl
for r in 1 2 3; do for v in ls1 ls0; do
  LAB_SHAPE=siglip_fc1 tools/lab/gemm_lab_$v w4p_bt 5 | sed "s/^{/{\"build\": \"$v\", /" >> gpurun_out/lab_gelu.jsonl
  LAB_SHAPE=whisper_fc1 tools/lab/gemm_lab_$v w4p_be 5 | sed "s/^{/{\"build\": \"$v\", /" >> gpurun_out/lab_gelu.jsonl
  LAB_SHAPE=mm_gateup tools/lab/gemm_lab_$v w4p_geglu 3 | sed "s/^{/{\"build\": \"$v\", /" >> gpurun_out/lab_gelu.jsonl
done; done
python - <<'PY'
import json, collections
acc=collections.defaultdict(list); cs=collections.defaultdict(set)
for l in open("gpurun_out/lab_gelu.jsonl"):
    if l.startswith("{"):
        d=json.loads(l); acc[(d["shape"],d["variant"],d["build"])].append(round(d["tflops"])); cs[(d["shape"],d["variant"])].add(d["checksum"])
for k in sorted(acc): print(k, acc[k])
print({k: len(v) for k,v in cs.items()}, "(1 = both builds give the same checksum)")
PY
