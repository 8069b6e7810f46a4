"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into HBM bytes per GEMM launch.
FETCH_SIZE/WRITE_SIZE are in KiB-ish units of 1024 B? (rocprofv3 reports kilobytes); on gfx950 FETCH_SIZE under-reports
wide coalesced streaming reads by exactly 2x (MI355X_MICROARCH.md §HBM) -> corrected here.
usage: traffic_from_pmc.py fetch_counter_collection.csv write_counter_collection.csv out.json"""
import csv, json, sys
def agg(path, counter, pat="gemm_kernel"):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and pat in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); n += 1
    return tot, n
f, nf = agg(sys.argv[1], "FETCH_SIZE")
w, nw = agg(sys.argv[2], "WRITE_SIZE")
out = {"kernel_family": "gemm_kernel", "launches_fetch_pass": nf, "launches_write_pass": nw,
       "fetch_bytes_per_launch_raw": f * 1024 / max(nf, 1), "fetch_bytes_per_launch_corrected_x2": 2 * f * 1024 / max(nf, 1),
       "write_bytes_per_launch": w * 1024 / max(nw, 1),
       "hbm_bytes_per_launch": (2 * f + w) * 1024 / max(nf, 1),
       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --no-cpu-baseline`; "
               "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B)"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
