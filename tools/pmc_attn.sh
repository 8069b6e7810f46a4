#!/bin/bash
# SQ stall accounting of the encoder-attention kernel (two PMC passes over tools/ab_attn.py on the shipped library): where the wave-cycles
# of attn_self_rm_kernel go — parked (s_waitcnt / barrier), issue-stalled, or issuing — and the LDS conflict share.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out
CMD="python $GRAFT_REPO_ROOT/tools/ab_attn.py $GRAFT_REPO_ROOT/${VIDI_PMC_LIB:-vidi_amd/libvidi_hip.so} --frames 360"      # VIDI_PMC_LIB: a variant library (tools/build_variant.sh)
(cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_attn1 -o a -- $CMD > /dev/null 2> $OUT/pmc_attn1.err)
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_attn2 -o a -- $CMD > /dev/null 2> $OUT/pmc_attn2.err)
python - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.Counter()); n = collections.Counter()
for d in ("pmc_attn1", "pmc_attn2"):
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "attn_self" in k:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
out = {}
for k, c in acc.items():
    wc = c["SQ_WAVE_CYCLES"] or 1
    out[k] = {"launches": n[(k, "SQ_WAVE_CYCLES")], **{x: c[x] for x in sorted(c)},
              "share_of_wave_cycles": {x: round(c[x] / wc, 4) for x in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS")},
              "lds_conflict_share": round(c["SQ_LDS_BANK_CONFLICT"] / max(1, c["SQ_LDS_IDX_ACTIVE"]), 4)}
json.dump(out, open("gpurun_out/pmc_attn.json", "w"), indent=1)
for k, v in out.items(): print(k, v["launches"], v["share_of_wave_cycles"], "lds conflicts", v["lds_conflict_share"], "insts VALU/LDS/SALU per launch", [round(v.get(x, 0) / max(1, v["launches"])) for x in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU")])
PY
rm -rf $OUT/pmc_attn1 $OUT/pmc_attn2
