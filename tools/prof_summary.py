"""Summarise a rocprofv3 --kernel-trace CSV: short kernel names, calls, total/avg µs, share.

    python tools/prof_summary.py <kernel_trace.csv> [--grid-filter train|sample] [--md]
"""
import csv
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    if name.startswith("at::native") or "at::native" in name[:40]:
        m = re.search(r"(normal_kernel|random_from_to|uniform_kernel|FillFunctor|direct_copy|MulFunctor|arange|add)", name)
        return "torch::" + (m.group(1) if m else name[:40])
    return re.sub(r"\(.*$", "", name)


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    agg = defaultdict(lambda: [0, 0.0, set()])
    for r in rows:
        n = short(r["Kernel_Name"])
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        grid = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
        key = (n, grid, int(r.get("Grid_Size_Y", 1) or 1))
        agg[key][0] += 1
        agg[key][1] += dur
    tot = sum(v[1] for v in agg.values())
    print(f"{'kernel':48s} {'wgs':>6s} {'gy':>3s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'share':>6s}")
    for (n, grid, gy), (c, d, _) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n[:48]:48s} {grid:6d} {gy:3d} {c:6d} {d:11.1f} {d / c:9.1f} {100 * d / tot:5.1f}%")
    print(f"total kernel time {tot / 1e3:.2f} ms over {len(rows)} dispatches")


if __name__ == "__main__":
    main()
