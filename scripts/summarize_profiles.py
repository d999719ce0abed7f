#!/usr/bin/env python
"""Condense rocprofv3 outputs (kernel stats + FETCH_SIZE / WRITE_SIZE counter passes) into one small JSON/CSV
summary per kernel family.  Usage: python scripts/summarize_profiles.py <gpurun_out dir>."""
import csv
import glob
import json
import os
import re
import sys

out = sys.argv[1]


def family(name):
    m = re.match(r"(?:void )?(edmp::[a-z_0-9]+)", name)
    return m.group(1) if m else name.split("(")[0][:60]


summary = {}
stats = glob.glob(os.path.join(out, "prof", "*kernel_stats.csv"))
if stats:
    for r in csv.DictReader(open(stats[0])):
        f = family(r["Name"])
        s = summary.setdefault(f, dict(calls=0, total_ns=0))
        s["calls"] += int(r["Calls"])
        s["total_ns"] += int(r["TotalDurationNs"])
for key, sub, col in (("fetch_kb", "pmc_fetch", "FETCH_SIZE"), ("write_kb", "pmc_write", "WRITE_SIZE")):
    files = glob.glob(os.path.join(out, sub, "*counter_collection.csv"))
    if not files:
        continue
    for r in csv.DictReader(open(files[0])):
        if r.get("Counter_Name") != col:
            continue
        f = family(r["Kernel_Name"])
        s = summary.setdefault(f, dict(calls=0, total_ns=0))
        s[key] = s.get(key, 0.0) + float(r["Counter_Value"])
        s[key + "_dispatches"] = s.get(key + "_dispatches", 0) + 1
tot = sum(s["total_ns"] for s in summary.values()) or 1
rows = []
for f, s in sorted(summary.items(), key=lambda kv: -kv[1]["total_ns"]):
    row = dict(kernel=f, calls=s["calls"], total_ms=s["total_ns"] / 1e6, avg_us=s["total_ns"] / 1e3 / max(s["calls"], 1), pct=100.0 * s["total_ns"] / tot)
    if "fetch_kb" in s:  # gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM) -> corrected
        row["fetch_MB_per_launch_corrected"] = 2.0 * s["fetch_kb"] * 1024 / 1e6 / s["fetch_kb_dispatches"]
    if "write_kb" in s:
        row["write_MB_per_launch"] = s["write_kb"] * 1024 / 1e6 / s["write_kb_dispatches"]
    rows.append(row)
json.dump(rows, open(os.path.join(out, "profile_summary.json"), "w"), indent=1)
for r in rows[:14]:
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()})
