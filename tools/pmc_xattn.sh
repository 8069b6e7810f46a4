#!/bin/bash
# SQ stall accounting of the many-row cross-attention kernel (two PMC passes over tools/bench_xattn.py): where the wave-cycles go.
# usage: bash tools/pmc_xattn.sh [Lq=304] [zsplit=6]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
LQ=${1:-304}; ZS=${2:-6}
CMD="python $GRAFT_REPO_ROOT/tools/bench_xattn.py --keys 90000 --lq $LQ --iters 4 --zsplit $ZS"
(cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_x1 -o a -- $CMD > /dev/null 2> $OUT/pmc_x1.err)
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_x2 -o a -- $CMD > /dev/null 2> $OUT/pmc_x2.err)
(cd /tmp && rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_CYCLES --output-format csv -d $OUT/pmc_x3 -o a -- $CMD > /dev/null 2> $OUT/pmc_x3.err)
python - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.Counter()); n = collections.Counter()
for d in ("pmc_x1", "pmc_x2", "pmc_x3"):
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "attn_cross" in k:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
out = {}
for k, c in acc.items():
    wc = c["SQ_WAVE_CYCLES"] or 1
    L = max(1, n[(k, "SQ_WAVE_CYCLES")])
    out[k] = {"launches": L, **{x: c[x] / L for x in sorted(c)},
              "share_of_wave_cycles": {x: round(c[x] / wc, 4) for x in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS")},
              "lds_conflict_share": round(c["SQ_LDS_BANK_CONFLICT"] / max(1, c["SQ_LDS_IDX_ACTIVE"]), 4)}
json.dump(out, open("gpurun_out/pmc_xattn.json", "w"), indent=1)
for k, v in out.items():
    print(k, json.dumps({a: (round(b) if isinstance(b, float) else b) for a, b in v.items()}))
PY
rm -rf $OUT/pmc_x1 $OUT/pmc_x2 $OUT/pmc_x3
