# LayerNorm statistics in the consumer's K loop (Epi::lnf == 2) against the statistics input (lnf == 1) on SigLIP's fc1 shape and the
# q|k|v shape: Y agreement, TFLOP/s of the whole kernel and of the K loop alone (noepi), cycle stamps.  Go / no-go for the product change:
# the K loop must not lose more than ~2 % (the producers' statistics epilogues and ln_finalize are worth ~1.5 % of the prefill).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/lab_lnstats.jsonl
: > $OUT
for r in 1 2 3; do for sh in siglip_fc1 siglip_qkv whisper_fc1; do
  LAB_SHAPE=$sh timeout 300 tools/lab/gemm_lab lnstats 5 | grep '^{' | sed "s/^{/{\"round\": $r, /" >> $OUT
done; done
python - <<'PY'
import json, collections
acc = collections.defaultdict(list)
for l in open("gpurun_out/lab_lnstats.jsonl"):
    d = json.loads(l)
    if "check" in d:
        if d["round"] == 1: print("check", d)
    elif "tflops" in d:
        acc[(d["shape"], d["variant"])].append(round(d["tflops"]))
    elif d["round"] == 1:
        print("stamps", d["shape"], d["variant"], "K loop", round(d["cycles_1"]), "epilogue", round(d["cycles_3"]), "clock", d.get("eff_clock_GHz"))
for k in sorted(acc): print(k, acc[k])
PY
