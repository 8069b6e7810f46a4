"""Per-kernel MFMA-busy share of SIMD cycles from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass.

SQ_VALU_MFMA_BUSY_CYCLES is summed over all 1 024 SIMDs (32 cycles per 32x32x16, 16 per 16x16x32 bf16 MFMA, padding included);
GRBM_GUI_ACTIVE is summed over the 8 XCDs, so busy share = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024).
usage: mfma_busy_from_pmc.py counter_collection.csv out.json"""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    return re.sub(r"^void ", "", name)[:110]


def main():
    busy, act, n = collections.Counter(), collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(sys.argv[1])):
        k = short(r["Kernel_Name"])
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            busy[k] += float(r["Counter_Value"]); n[k] += 1
        elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            act[k] += float(r["Counter_Value"])
    out = {k: {"launches": n[k], "mfma_busy_frac_of_simd_cycles": busy[k] / (act[k] / 8 * 1024)} for k in busy if act[k] > 0 and busy[k] > 0}
    out = dict(sorted(out.items(), key=lambda kv: -busy[kv[0]]))
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    for k, v in list(out.items())[:14]:
        print(f"{v['mfma_busy_frac_of_simd_cycles']:.3f}  {v['launches']:5d}  {k}")


if __name__ == "__main__":
    main()
