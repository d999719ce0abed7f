#!/usr/bin/env python
"""Where the time BETWEEN kernels goes: reads a rocprofv3 --kernel-trace CSV (one bench.py call), orders the dispatches by start time and
aggregates the idle gap between consecutive kernels by (previous kernel -> next kernel).  python scripts/gap_analysis.py <dir or csv> [top = 25]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

src = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()


def short(n):
    n = re.sub(r"\(edmp::WideKind\)|\(edmp::LevelMode\)|\(edmp::GuideMode\)|void |edmp::", "", n)
    m = re.match(r"([A-Za-z_0-9:]+(<[^>]*>)?)", n)
    return (m.group(1) if m else n)[:60]


gaps = defaultdict(lambda: [0, 0])
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
idle = 0
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = max(s1 - e0, 0)
    if g > 200000:  # > 0.2 ms: between calls / host phases, listed separately
        gaps[("<long pause>", short(n1))][0] += g
        gaps[("<long pause>", short(n1))][1] += 1
        continue
    idle += g
    k = (short(n0), short(n1))
    gaps[k][0] += g
    gaps[k][1] += 1
print(f"{len(rows)} dispatches, span {span / 1e6:.2f} ms, kernels busy {busy / 1e6:.2f} ms, short gaps {idle / 1e6:.2f} ms ({100.0 * idle / max(busy, 1):.2f} % of busy)")
print("| previous -> next | count | total gap us | avg gap us |\n|---|---:|---:|---:|")
for (a, b), (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"| `{a}` -> `{b}` | {c} | {t / 1e3:.1f} | {t / 1e3 / c:.2f} |")
