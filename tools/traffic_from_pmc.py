"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into fabric-side bytes per GEMM launch.

rocprofv3 reports both counters in KiB.  On gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by exactly 2x
(MI355X_MICROARCH.md, HBM section) -> corrected here.  Both counters sit on the L2's memory side: Infinity-Cache hits are counted,
so the sum is an UPPER bound on HBM traffic (same guide).  The output is keyed by the digest of the GEMM sources it was collected
on; bench.py refuses to attach it to any other build.

usage: traffic_from_pmc.py fetch_counter_collection.csv write_counter_collection.csv out.json [frames] [preset] [command]"""
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PAT = re.compile(r"gemm_w4n?_kernel|gemm_kernel<|gemm_kernel\(|skinny_kernel<")     # one kernel per vidi_gemm* call (skinny_reduce_kernel rides along uncounted: < 1 % of its launch's bytes)


def agg(path, counter):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and PAT.search(r["Kernel_Name"]):
            tot += float(r["Counter_Value"]); n += 1
    return tot, n


def per_kernel(fetch_csv, write_csv):
    """fabric-side bytes per launch of every GEMM instantiation (the epilogue template arguments tell the shapes apart): FETCH_SIZE x2 and
    WRITE_SIZE, so the over-fetch can be read per shape instead of family-wide"""
    import collections
    f, w, nf = collections.Counter(), collections.Counter(), collections.Counter()
    for path, counter, acc in ((fetch_csv, "FETCH_SIZE", f), (write_csv, "WRITE_SIZE", w)):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and PAT.search(r["Kernel_Name"]):
                k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:120]
                acc[k] += float(r["Counter_Value"]) * 1024
                if counter == "FETCH_SIZE":
                    nf[k] += 1
    return {k: {"launches": nf[k], "fetch_x2_bytes_per_launch": 2 * f[k] / nf[k], "write_bytes_per_launch": w[k] / nf[k]} for k in sorted(nf, key=lambda k: -f[k])}


def main():
    from vidi_amd.build import source_digest
    f, nf = agg(sys.argv[1], "FETCH_SIZE")
    w, nw = agg(sys.argv[2], "WRITE_SIZE")
    out = {"kernel_family": "gemm_w4_kernel + gemm_kernel", "launches_fetch_pass": nf, "launches_write_pass": nw,
           "fetch_bytes_per_launch_raw": f * 1024 / max(nf, 1), "fetch_bytes_per_launch_corrected_x2": 2 * f * 1024 / max(nf, 1),
           "write_bytes_per_launch": w * 1024 / max(nw, 1),
           "hbm_bytes_per_launch": (2 * f * 1024 / max(nf, 1)) + (w * 1024 / max(nw, 1)),
           "gemm_source_digest": source_digest("gemm"),
           "frames": int(sys.argv[4]) if len(sys.argv) > 4 else 3600, "preset": sys.argv[5] if len(sys.argv) > 5 else "vidi15_9b",
           "command": sys.argv[6] if len(sys.argv) > 6 else "python bench.py --no-cpu-baseline --no-kernel-timer --no-preproc --steps 1 --warmup 0 --decode-steps 2",
           "file": "profiles/traffic.json",
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 "
                   "counts 128-B requests at 64 B); fabric-side counters: Infinity-Cache hits included (upper bound on HBM bytes)"}
    out["per_kernel"] = per_kernel(sys.argv[1], sys.argv[2])
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "per_kernel"}))
    for k, v in out["per_kernel"].items():
        print(f"{v['launches']:5d} launches  fetch x2 {v['fetch_x2_bytes_per_launch'] / 1e9:7.2f} GB  write {v['write_bytes_per_launch'] / 1e9:6.2f} GB  {k}")


if __name__ == "__main__":
    main()
