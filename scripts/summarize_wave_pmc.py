#!/usr/bin/env python3
"""Per-kernel wave-cycle breakdown from a rocprofv3 PMC pass over scripts/pmc_unet_forward.py with
  --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS ...
and/or --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE.  WAIT_ANY = parked at s_waitcnt / s_barrier, WAIT_INST_ANY = issue
stall behind a busy pipe (for these kernels: the MFMA pipe), ACTIVE_INST_ANY = issuing; the three add up to the wave
cycles (MI355X_MICROARCH.md).  usage: summarize_wave_pmc.py <counter_collection.csv> [...more csv] [-o out.md]"""
import collections
import csv
import sys

args = [a for a in sys.argv[1:] if a != "-o"]
out_path = None
if "-o" in sys.argv:
    out_path = sys.argv[sys.argv.index("-o") + 1]
    args.remove(out_path)
agg = collections.OrderedDict()
for path in args:
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"].split("(")[0].replace("edmp::", "").replace("void ", "")
        agg.setdefault(n, collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
lines = ["| kernel | parked (s_waitcnt / barrier) | issue-stalled (MFMA pipe busy) | issuing | of which VALU | of which LDS | LDS bank-conflict cycles / LDS active |",
         "|---|---:|---:|---:|---:|---:|---:|"]
for n, a in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    w = a.get("SQ_WAVE_CYCLES", 0)
    if w == 0 or "mfma" not in n and "rcb" not in n:
        continue
    f = lambda k: f"{100 * a.get(k, 0) / w:.0f} %"  # noqa: E731
    bc = f"{100 * a['SQ_LDS_BANK_CONFLICT'] / a['SQ_LDS_IDX_ACTIVE']:.0f} %" if a.get("SQ_LDS_IDX_ACTIVE") else "-"
    lines.append(f"| `{n}` | {f('SQ_WAIT_ANY')} | {f('SQ_WAIT_INST_ANY')} | {f('SQ_ACTIVE_INST_ANY')} | {f('SQ_ACTIVE_INST_VALU')} | {f('SQ_ACTIVE_INST_LDS')} | {bc} |")
txt = "\n".join(lines) + "\n"
if out_path:
    open(out_path, "w").write(txt)
print(txt)
