#!/usr/bin/env python3
"""Summarise a VK_CONV_LOG file: per distinct conv shape -> launches, total ms, TFLOP/s."""
import collections
import sys
rows = collections.OrderedDict()
for line in open(sys.argv[1]):
    b, M, co, ci, k, s, ms, tf = line.split()
    key = (int(b), int(M), int(co), int(ci), int(k), int(s))
    r = rows.setdefault(key, [0, 0.0])
    r[0] += 1
    r[1] += float(ms)
tot = sum(r[1] for r in rows.values())
print(f"{'bkt':>3} {'M':>9} {'cout':>5} {'cin':>5} {'k':>2} {'s':>1} {'n':>6} {'ms_tot':>9} {'share':>6} {'us/launch':>9} {'TFLOP/s':>8}")
for (b, M, co, ci, k, s), (n, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    fl = 2.0 * M * co * ci * k * n
    print(f"{b:>3} {M:>9} {co:>5} {ci:>5} {k:>2} {s:>1} {n:>6} {ms:>9.2f} {ms / tot:>6.1%} {ms / n * 1e3:>9.1f} {fl / (ms * 1e-3) / 1e12:>8.1f}")
print(f"total {tot:.2f} ms")
