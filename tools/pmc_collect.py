#!/usr/bin/env python3
"""Fold the rocprofv3 passes of tools/roundend2.sh into one JSON per kernel (name + grid):
    python tools/pmc_collect.py <dir with pass sub-directories> <out.json>
FETCH_SIZE is doubled (gfx950: the counter tallies 128-B requests at 64 B — MI355X_MICROARCH.md, HBM section);
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main():
    root, out = sys.argv[1], sys.argv[2]
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0) // max(1, int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)) or 1))
            agg[(short(r["Kernel_Name"]), grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            grid = 1
            for ax in "XYZ":
                grid *= max(1, int(r.get("Grid_Size_" + ax, 1) or 1)) // max(1, int(r.get("Workgroup_Size_" + ax, 1) or 1))
            agg[(short(r["Kernel_Name"]), grid)]["duration_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    res = []
    for (name, grid), d in sorted(agg.items()):
        if not any(k in name for k in ("conv_s16", "wgrad2_s16", "gemm_nt_bf16")):
            continue
        e = {"kernel": name, "workgroups": grid}
        for c, v in d.items():
            v = v[1:] if len(v) > 2 else v          # drop the first (cold) launch
            e[c] = sum(v) / len(v)
            e[c + "_n"] = len(v)
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = int(2 * e["FETCH_SIZE"] * 1024 + e["WRITE_SIZE"] * 1024)
        if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e:
            e["l2_hit_rate"] = round(e["TCC_HIT_sum"] / max(1.0, e["TCC_HIT_sum"] + e["TCC_MISS_sum"]), 4)
        if "SQ_INSTS_VALU" in e and "SQ_INSTS_MFMA" in e:
            e["valu_per_mfma"] = round(e["SQ_INSTS_VALU"] / max(1.0, e["SQ_INSTS_MFMA"]), 2)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "SQ_BUSY_CU_CYCLES" in e:
            e["mfma_busy_frac_of_cu_busy"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1.0, 4.0 * e["SQ_BUSY_CU_CYCLES"]), 4)
        if "duration_ns" in e and "hbm_bytes_per_launch" in e:
            e["hbm_gbs"] = round(e["hbm_bytes_per_launch"] / e["duration_ns"], 1)
        res.append(e)
    json.dump({"correction": "FETCH_SIZE x2 (gfx950), KB -> bytes; first launch of each kernel dropped", "kernels": res}, open(out, "w"), indent=1)
    for e in res:
        print(e["kernel"][:46], e["workgroups"], {k: (round(v, 1) if isinstance(v, float) else v) for k, v in e.items()
                                                   if k in ("duration_ns", "hbm_bytes_per_launch", "l2_hit_rate", "valu_per_mfma", "mfma_busy_frac_of_cu_busy", "hbm_gbs")})


if __name__ == "__main__":
    main()
