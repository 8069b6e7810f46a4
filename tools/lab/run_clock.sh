# tile order x effective clock x fabric traffic: cycle stamps per order / group size on the widest GEMMs, then the
# same variants under rocprofv3 --pmc FETCH_SIZE (one pass per variant: the lab binary runs 7 launches of it)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/lab_clock.jsonl
: > $OUT
for sh in mm_gateup siglip_fc1 mm_down; do
  LAB_SHAPE=$sh timeout 300 tools/lab/gemm_lab clock 3 | grep '^{' >> $OUT
  LAB_SHAPE=$sh timeout 300 tools/lab/gemm_lab orders 5 | grep '^{' >> $OUT
done
export TMPDIR=/tmp
for v in w4p_g1 w4p_o1 w4p_g16; do
  rm -rf /tmp/pmc_$v
  (cd /tmp && LAB_SHAPE=mm_gateup timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_$v -o f -- $GRAFT_REPO_ROOT/tools/lab/gemm_lab $v 2 > /dev/null 2>&1)
  f=$(find /tmp/pmc_$v -name '*counter_collection.csv' | head -1)
  python - "$f" $v <<'PY' >> $OUT
import csv, sys, json
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm_w4_kernel" in r.get("Kernel_Name", "") and r.get("Counter_Name") == "FETCH_SIZE"]
vals = [float(r["Counter_Value"]) for r in rows]
print(json.dumps({"pmc": "FETCH_SIZE", "variant": sys.argv[2], "shape": "mm_gateup", "launches": len(vals), "fetch_kb_per_launch_raw": sum(vals) / max(1, len(vals)),
                  "fetch_GB_per_launch_x2": 2 * sum(vals) / max(1, len(vals)) * 1024 / 1e9}))
PY
done
cat $OUT | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'eff_clock_GHz' in d: print(d['shape'], d['variant'], 'order', d['order'], 'group', d['group_m'], 'clock', d['eff_clock_GHz'], 'cyc/tile', round((d['cycles_0']+d['cycles_1']+d['cycles_2']+d['cycles_3'])*d['blocks']/d['tiles']))
    elif 'tflops' in d: print(d['shape'], d['variant'], d['tflops'], d['vs_ref'])
    else: print(d)
"
