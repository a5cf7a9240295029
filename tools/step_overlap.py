#!/usr/bin/env python3
"""Timeline of a train step whose launches OVERLAP (two queues), from a rocprofv3 --kernel-trace CSV: steps are cut at
draw_q_sample_kernel dispatches; every launch is keyed by (kernel name, occurrence inside the step) and reported with its
average start offset, duration and end, sorted by start; `busy` = union of kernel intervals, `sum` = their total.

    python tools/step_overlap.py <kernel_trace.csv> [--skip 5]
"""
import csv
import re
import sys
from collections import Counter, defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main():
    path = sys.argv[1]
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 5
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    ev = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
           int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r.get("Queue_Id", "")) for r in rows]
    starts = [i for i, e in enumerate(ev) if e[0].startswith("draw_q_sample_kernel")]
    steps = [ev[a:b] for a, b in zip(starts[:-1], starts[1:])]
    n = Counter(len(s) for s in steps).most_common(1)[0][0]
    steps = [s for s in steps if len(s) == n]
    # (the bench also runs the step at other batch sizes — same launch count: keep the steps of the most common workgroup signature)
    sig = Counter(tuple(e[3] for e in s) for s in steps).most_common(1)[0][0]
    steps = [s for s in steps if tuple(e[3] for e in s) == sig][skip:]
    acc = defaultdict(lambda: [0.0, 0.0, 0, 0, ""])
    busy = tot = span = 0.0
    for s in steps:
        t0 = s[0][1]
        occ = Counter()
        iv = []
        for name, a, b, wg, q in s:
            k = (name, occ[name])
            occ[name] += 1
            r = acc[k]
            r[0] += (a - t0) / 1e3
            r[1] += (b - a) / 1e3
            r[2] += 1
            r[3] = wg
            r[4] = q
            iv.append((a, b))
            tot += (b - a) / 1e3
        iv.sort()
        ca, cb = iv[0]
        for a, b in iv[1:]:
            if a > cb:
                busy += (cb - ca) / 1e3
                ca, cb = a, b
            else:
                cb = max(cb, b)
        busy += (cb - ca) / 1e3
        span += (max(b for _, b in iv) - t0) / 1e3
    ns = len(steps)
    print(f"{ns} steps of {n} launches")
    print(f"{'start':>8s} {'dur':>7s} {'end':>8s} {'wgs':>6s} {'queue':>6s}  kernel")
    for (name, o), r in sorted(acc.items(), key=lambda kv: kv[1][0] / kv[1][2]):
        a, d = r[0] / r[2], r[1] / r[2]
        print(f"{a:8.1f} {d:7.1f} {a + d:8.1f} {r[3]:6d} {r[4]:>6s}  {name[:60]} #{o}")
    period = (steps[-1][0][1] - steps[0][0][1]) / max(1, ns - 1) / 1e3
    print(f"sum of kernel durations {tot / ns:.1f} us, union of their intervals {busy / ns:.1f} us, first start -> last end {span / ns:.1f} us, step period {period:.1f} us")


if __name__ == "__main__":
    main()
