# same-box A/B of the statistics epilogue's row reductions (DPP on the add vs the builtin form), alternating builds, checksums of Y and of the partial sums
cd $GRAFT_REPO_ROOT
: > gpurun_out/lab_stats.jsonl
for r in 1 2 3; do for v in dpp1 dpp0; do for sh in siglip_o siglip_fc2; do
  LAB_SHAPE=$sh tools/lab/gemm_lab_$v w4p_br 5 | sed "s/^{/{\"build\": \"$v\", /" >> gpurun_out/lab_stats.jsonl
  LAB_SHAPE=$sh tools/lab/gemm_lab_$v w4p_brs 5 | sed "s/^{/{\"build\": \"$v\", /" >> gpurun_out/lab_stats.jsonl
done; done; done
python - <<'PY'
import json, collections
acc=collections.defaultdict(list); cs=collections.defaultdict(set)
for l in open("gpurun_out/lab_stats.jsonl"):
    if l.startswith("{"):
        d=json.loads(l); acc[(d["shape"],d["variant"],d["build"])].append(round(d["tflops"])); cs[(d["shape"],d["variant"])].add(d["checksum"])
for k in sorted(acc): print(k, acc[k])
print({k: len(v) for k,v in cs.items()}, "(1 = both builds give the same checksum)")
PY
