#!/usr/bin/env python
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_unet_forward.py -> per-kernel HBM traffic summary.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane)
coalesced read stream (MI355X_MICROARCH.md, HBM section), so the read side is doubled; WRITE_SIZE is used as is.
Usage: python scripts/summarize_pmc.py <dir with pmc_FETCH_SIZE/ pmc_WRITE_SIZE/> <out.json> [n_forwards]"""
import collections
import csv
import glob
import json
import os
import re
import sys

src, out = sys.argv[1], sys.argv[2]
nfwd = int(sys.argv[3]) if len(sys.argv) > 3 else 3


def fam(n):
    m = re.match(r"(?:void )?(edmp::[a-z_0-9]+(?:<[^>]*>)?)", n)
    return m.group(1) if m else n[:40]


res = collections.defaultdict(lambda: dict(launches=0, fetch_kib=0.0, write_kib=0.0, wl=0))
for r in csv.DictReader(open(glob.glob(os.path.join(src, "pmc_FETCH_SIZE", "*counter_collection.csv"))[0])):
    if r["Counter_Name"] == "FETCH_SIZE":
        k = fam(r["Kernel_Name"])
        res[k]["launches"] += 1
        res[k]["fetch_kib"] += float(r["Counter_Value"])
for r in csv.DictReader(open(glob.glob(os.path.join(src, "pmc_WRITE_SIZE", "*counter_collection.csv"))[0])):
    if r["Counter_Name"] == "WRITE_SIZE":
        k = fam(r["Kernel_Name"])
        res[k]["wl"] += 1
        res[k]["write_kib"] += float(r["Counter_Value"])
rows, conv_bytes, conv_launches = [], 0.0, 0
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["fetch_kib"]):
    fetch = 2.0 * v["fetch_kib"] * 1024
    write = v["write_kib"] * 1024
    rows.append(dict(kernel=k, launches=v["launches"], fetch_MB_per_launch_corrected=fetch / 1e6 / max(v["launches"], 1),
                     write_MB_per_launch=write / 1e6 / max(v["wl"], 1)))
    if "conv" in k or "rcb" in k or "level" in k:
        conv_bytes += fetch + write
        conv_launches += v["launches"]
summary = dict(
    source="rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python scripts/pmc_unet_forward.py (B=1024, 3 forwards)",
    correction="read side = 2 x FETCH_SIZE KiB (gfx950 under-report of 16 B/lane streams), write side = WRITE_SIZE KiB",
    conv_family=dict(launches_per_forward=conv_launches // nfwd, hbm_MB_per_forward=conv_bytes / 1e6 / nfwd,
                     hbm_MB_per_launch=conv_bytes / 1e6 / max(conv_launches, 1), hbm_bytes_per_traj_step=conv_bytes / nfwd / 1024),
    kernels=rows,
)
json.dump(summary, open(out, "w"), indent=1)
print(json.dumps(summary["conv_family"]))
