import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = {}
for r in rows:
    if pat not in r["Kernel_Name"]:
        continue
    k = (r["Kernel_Name"][:60], r["Counter_Name"])
    agg.setdefault(k, []).append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k, "avg=%.4g" % (sum(v) / len(v)), "n=%d" % len(v))
r = rows[0]
print({k: r[k] for k in r if any(s in k for s in ("GPR", "LDS", "Scratch", "Workgroup", "Grid"))})
