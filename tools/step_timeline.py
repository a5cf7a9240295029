#!/usr/bin/env python3
"""In-pipeline timeline of the train step from a rocprofv3 --kernel-trace CSV: for every launch position between
two consecutive draw_q_sample_kernel dispatches, the average duration and the average idle gap before it.

    python tools/step_timeline.py <kernel_trace.csv> [--skip 5]
"""
import csv
import re
import sys
from collections import Counter


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main():
    path = sys.argv[1]
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 5
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    ev = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
           int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) for r in rows]
    starts = [i for i, e in enumerate(ev) if e[0].startswith("draw_q_sample_kernel")]
    steps = [ev[a:b] for a, b in zip(starts[:-1], starts[1:])]
    if not steps:
        print("no draw_q_sample_kernel dispatches found")
        return
    n = Counter(len(s) for s in steps).most_common(1)[0][0]
    steps = [s for s in steps if len(s) == n]
    sig = Counter(tuple(e[3] for e in s) for s in steps).most_common(1)[0][0]   # (one batch size: the most common workgroup signature)
    steps = [s for s in steps if tuple(e[3] for e in s) == sig][skip:]
    print(f"{len(steps)} steps of {n} launches")
    tot_d = tot_g = 0.0
    print(f"{'#':>3s} {'avg_us':>8s} {'gap_us':>7s} {'wgs':>6s}  kernel")
    for k in range(n):
        d = sum((s[k][2] - s[k][1]) for s in steps) / len(steps) / 1e3
        g = sum((s[k][1] - s[k - 1][2]) for s in steps) / len(steps) / 1e3 if k else 0.0
        tot_d += d
        tot_g += g
        print(f"{k:3d} {d:8.2f} {g:7.2f} {steps[0][k][3]:6d}  {steps[0][k][0][:70]}")
    span = sum((s[-1][2] - s[0][1]) for s in steps) / len(steps) / 1e3
    period = (steps[-1][0][1] - steps[0][0][1]) / max(1, len(steps) - 1) / 1e3
    print(f"sum of kernel durations {tot_d:.1f} us, gaps inside a step {tot_g:.1f} us, first start -> last end {span:.1f} us, step period {period:.1f} us")


if __name__ == "__main__":
    main()
