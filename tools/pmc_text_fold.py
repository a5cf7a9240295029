#!/usr/bin/env python3
"""Fold tools/pmc_text.sh's passes per (kernel, grid): average duration, HBM bytes (FETCH_SIZE x2 on gfx950, KB -> bytes,
MI355X_MICROARCH.md HBM section), GB/s.  The first step of each pass (eager, cold) is dropped by skipping the first third of
every kernel's launches."""
import csv, glob, json, re, sys
from collections import defaultdict
root, out = sys.argv[1], sys.argv[2]
def short(n):
    n = n.replace("(anonymous namespace)::", ""); n = re.sub(r"^void ", "", n); return re.sub(r"\(.*$", "", n)
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(s in r["Kernel_Name"] for s in ("at::", "rocclr", "rocprim", "__amd_")): continue
        wg = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
        agg[(short(r["Kernel_Name"]), wg)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(s in r["Kernel_Name"] for s in ("at::", "rocclr", "rocprim", "__amd_")): continue
        wg = 1
        for ax in "XYZ": wg *= max(1, int(r["Grid_Size_" + ax])) // max(1, int(r["Workgroup_Size_" + ax]))
        agg[(short(r["Kernel_Name"]), wg)]["duration_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res = []
for (k, wg), d in agg.items():
    e = {"kernel": k, "workgroups": wg}
    for c, v in d.items():
        v = v[len(v) // 3:]
        e[c] = sum(v) / len(v); e["launches_per_step"] = len(d[c]) // 3
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e and "duration_ns" in e:
        e["hbm_bytes_per_launch"] = int(2 * e["FETCH_SIZE"] * 1024 + e["WRITE_SIZE"] * 1024)
        e["hbm_gbs"] = round(e["hbm_bytes_per_launch"] / e["duration_ns"], 1)
        e["us_per_step"] = round(e["duration_ns"] * e["launches_per_step"] / 1e3, 1)
        res.append(e)
res.sort(key=lambda e: -e["us_per_step"])
json.dump({"correction": "FETCH_SIZE x2 (gfx950), KB -> bytes; first of three steps dropped", "kernels": res}, open(out, "w"), indent=1)
for e in res[:24]:
    print(f'{e["kernel"][:48]:48s} wg {e["workgroups"]:6d} x{e["launches_per_step"]:3d}  {e["duration_ns"]/1e3:7.1f} us  {e["hbm_bytes_per_launch"]/1e6:7.1f} MB  {e["hbm_gbs"]:7.1f} GB/s')
