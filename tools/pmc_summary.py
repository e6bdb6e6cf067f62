#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per (kernel, counter).  usage: pmc_summary.py <dir> [kernel-substr]"""
import collections
import csv
import glob
import sys
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        a = acc[(k[:60], r["Counter_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(acc.items()):
    print(f"{k:60s} {c:32s} n={n:6d} sum={v:.6g} per_dispatch={v / n:.6g}")
