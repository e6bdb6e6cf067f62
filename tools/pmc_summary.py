#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per (kernel, counter).
usage: pmc_summary.py <dir> [kernel-substr]                 print per-kernel sums
       pmc_summary.py <dir> <kernel-substr> --json <out> [--name <kernel name for bench.py>] [--batch B --proposals R] [--min-workgroups N]
           write profiles/r01_pmc_traffic.json-style HBM bytes per launch of that kernel (FETCH_SIZE doubled on gfx950)"""
import collections
import csv
import glob
import json
import sys
acc = collections.defaultdict(lambda: [0, 0.0])
sub = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None
# --min-workgroups N: only dispatches of at least N workgroups (the launches that run alone: the res3/res4 half-batch launches of
# the two-stream backbone section have < 10000 workgroups and overlap each other)
MIN_WG = int(sys.argv[sys.argv.index("--min-workgroups") + 1]) if "--min-workgroups" in sys.argv else 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if sub and sub not in k:
            continue
        if MIN_WG and int(r.get("Grid_Size", 0)) // max(int(r.get("Workgroup_Size", 1)), 1) < MIN_WG:
            continue
        a = acc[(k[:60], r["Counter_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(acc.items()):
    print(f"{k:60s} {c:32s} n={n:6d} sum={v:.6g} per_dispatch={v / n:.6g}")
if "--json" in sys.argv:
    opt = {sys.argv[i]: sys.argv[i + 1] for i in range(3, len(sys.argv) - 1) if sys.argv[i].startswith("--")}
    opt.setdefault("--min-workgroups", "0")
    fetch = sum(v for (k, c), (n, v) in acc.items() if c == "FETCH_SIZE")
    write = sum(v for (k, c), (n, v) in acc.items() if c == "WRITE_SIZE")
    launches = max(n for (k, c), (n, v) in acc.items() if c == "FETCH_SIZE")
    out = {"batch": int(opt.get("--batch", 32)), "proposals": int(opt.get("--proposals", 300)), "kernel": opt.get("--name", sub),
           "kernel_symbol_filter": sub,
           "launches": launches, "FETCH_SIZE_KB_sum": fetch, "WRITE_SIZE_KB_sum": write,
           "correction": "FETCH_SIZE x2 on gfx950 (wide coalesced reads are tallied at half), WRITE_SIZE as read "
                         "(MI355X_MICROARCH.md, HBM section)",
           "hbm_bytes_per_launch": round((2 * fetch + write) * 1024 / launches),
           "min_workgroups": int(opt["--min-workgroups"]),
           # algorithmic bytes (input + output (+ residual) + weights once) of exactly the launches the filter selects, if given
           "alg_bytes_per_launch": int(opt["--alg-bytes"]) if "--alg-bytes" in opt else None,
           "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (two passes) -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline"}
    json.dump(out, open(opt["--json"], "w"), indent=1)
    print("wrote", opt["--json"], out["hbm_bytes_per_launch"])
