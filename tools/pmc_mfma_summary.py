#!/usr/bin/env python3
"""MFMA utilisation and effective clock per kernel from a rocprofv3 PMC pass
(`--kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`, its own run: tools/refresh_profiles.sh).

  busy fraction   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)
                    (counter_defs.yaml `MfmaUtil`: sum over SIMDs / (GUI-active cycles * SIMD_NUM); rocprofv3 reports
                     GRBM_GUI_ACTIVE summed over the 8 XCDs, MI355X_MICROARCH.md "DVFS give-back"; 1024 SIMDs = 256 CUs x 4)
  effective clock = GRBM_GUI_ACTIVE / 8 / kernel duration  (reads high on dispatches shorter than ~0.3 ms, same section)

usage: pmc_mfma_summary.py <dir with *counter_collection.csv [+ *kernel_trace.csv]> [--json out.json] [--top N] [--min-us T]
"""
import collections
import csv
import glob
import json
import re
import sys

d = sys.argv[1]
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 8
min_us = float(sys.argv[sys.argv.index("--min-us") + 1]) if "--min-us" in sys.argv else 0.0
dur = {}            # dispatch id -> ns (kernel trace of the same run, if the counter file carries no timestamps)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        try:
            dur[r.get("Dispatch_Id")] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        except Exception:
            pass
per = collections.defaultdict(dict)     # dispatch id -> {counter: value, "name":, "ns":}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        e = per[r["Dispatch_Id"]]
        e["name"] = r["Kernel_Name"]
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if "Start_Timestamp" in r and r.get("End_Timestamp"):
            try:
                e["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            except Exception:
                pass
        if "ns" not in e and r["Dispatch_Id"] in dur:
            e["ns"] = dur[r["Dispatch_Id"]]


def short(n):
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)[:70]


agg = collections.defaultdict(lambda: {"n": 0, "busy": 0.0, "gui": 0.0, "ns": 0.0})
for e in per.values():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in e or "GRBM_GUI_ACTIVE" not in e:
        continue
    if e.get("ns", 0) < min_us * 1e3:
        continue
    a = agg[short(e["name"])]
    a["n"] += 1
    a["busy"] += e["SQ_VALU_MFMA_BUSY_CYCLES"]
    a["gui"] += e["GRBM_GUI_ACTIVE"]
    a["ns"] += e.get("ns", 0)
rows = []
for k, a in agg.items():
    cyc = a["gui"] / 8.0
    rows.append({"kernel": k, "dispatches": a["n"], "gpu_ms": round(a["ns"] / 1e6, 3),
                 "mfma_busy_frac": round(a["busy"] / (cyc * 1024.0), 4) if cyc else None,
                 "effective_clock_ghz": round(cyc / a["ns"], 3) if a["ns"] else None})
rows.sort(key=lambda r: -r["gpu_ms"])
for r in rows[:top]:
    print(f"{r['kernel']:70s} n={r['dispatches']:4d} gpu_ms={r['gpu_ms']:9.3f} mfma_busy={r['mfma_busy_frac']} clock_ghz={r['effective_clock_ghz']}")
if "--json" in sys.argv:
    out = {"command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 1 --warmup 0 "
                      "--no-cpu-baseline --no-parity",
           "formulas": {"mfma_busy_frac": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)",
                        "effective_clock_ghz": "GRBM_GUI_ACTIVE / 8 / dispatch duration (profiled passes clock 2-5 % below un-profiled ones)"},
           "min_dispatch_us": min_us, "kernels": rows[:top]}
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
