#!/usr/bin/env python3
"""Idle time between consecutive kernels of the timed loop, from a rocprofv3 --kernel-trace CSV:
    python tools/kernel_gaps.py <kernel_trace.csv>
prints, for the k_step -> render and render -> k_step boundaries, the mean gap between one kernel's end and the next one's start."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
gaps = defaultdict(list)
short = lambda n: "k_step" if "k_step" in n else ("render" if "k_observe" in n and "codes" not in n else n[:24])
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    a, b = short(n0), short(n1)
    if {a, b} <= {"k_step", "render"} and a != b:
        gaps[(a, b)].append(s1 - e0)
for k, v in gaps.items():
    v = sorted(v)[len(v) // 10: -max(1, len(v) // 10)]  # trim the setup phases' outliers
    print(f"{k[0]:>7s} -> {k[1]:<7s}: {len(v)} boundaries, mean gap {sum(v) / len(v) / 1000:.2f} us, min {v[0] / 1000:.2f}, max {v[-1] / 1000:.2f}")
dur = defaultdict(list)
for s, e, n in rows:
    dur[short(n)].append(e - s)
for k in ("k_step", "render"):
    v = sorted(dur[k])[len(dur[k]) // 10: -max(1, len(dur[k]) // 10)]
    print(f"{k}: mean duration {sum(v) / len(v) / 1000:.2f} us over {len(v)} launches")
