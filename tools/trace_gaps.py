#!/usr/bin/env python3
"""GPU idle time inside the bench's steps from a rocprofv3 --kernel-trace CSV: union of the kernels' [start, end) intervals against
the span, and the largest gaps with the kernels on both sides.   usage: trace_gaps.py <dir with *kernel_trace.csv> [n_gaps]"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
# the timed region: from the first kernel after the longest gap (warm-up / parity passes end there) -- simply take the last 40 %
t_lo = rows[int(len(rows) * 0.6)][0]
rows = [r for r in rows if r[0] >= t_lo]
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e, gaps, last_name = 0, rows[0][0], rows[0][1], [], rows[0][2]
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e:
        last_name = n
busy += cur_e - cur_s
print(f"{len(rows)} kernels over {span / 1e6:.2f} ms: GPU busy {busy / 1e6:.2f} ms = {100 * busy / span:.2f} %, idle {100 * (1 - busy / span):.2f} % in {len(gaps)} gaps")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
agg = {}
for g, a, b in gaps:
    k = (a, b)
    agg[k] = (agg.get(k, (0, 0))[0] + g, agg.get(k, (0, 0))[1] + 1)
for (a, b), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:n]:
    print(f"  {g / 1e3:9.1f} us in {c:4d} gaps   {a}  ->  {b}")
