#!/usr/bin/env python3
"""Copy a rocprofv3 CSV keeping only this library's dispatches (torch's templated kernel names run to kilobytes per row)."""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
rd = csv.DictReader(open(src))
wr = csv.DictWriter(open(dst, "w", newline=""), fieldnames=rd.fieldnames)
wr.writeheader()
for r in rd:
    if not any(s in r["Kernel_Name"] for s in ("at::", "rocclr", "rocprim", "hipcub", "__amd_")):
        wr.writerow(r)
