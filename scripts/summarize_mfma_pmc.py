#!/usr/bin/env python3
"""Per-kernel MFMA-busy fraction from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass over
scripts/pmc_unet_forward.py (one UNet forward of the bench batch).

MfmaUtil as rocprofv3 defines it:  100 * sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * SIMD_NUM), SIMD_NUM =
256 CUs * 4.  The csv holds GRBM_GUI_ACTIVE summed over the 8 XCD instances, so max() = sum / 8 (checked against
the dispatch's End-Start timestamps, see the note the script prints).
SQ_VALU_MFMA_BUSY_CYCLES is 64 per v_mfma_f32_32x32x2_f32 (MI355X_MICROARCH.md cycle table), which the totals
reproduce: executed MFMAs/forward x 64 matches the sum to the tile-padding waste.  usage: summarize_mfma_pmc.py <counter_collection.csv> [out.md]"""
import csv
import sys
from collections import OrderedDict

SIMD_NUM = 256 * 4
XCDS = 8


def short(name: str) -> str:
    n = name.split("(")[0].replace("edmp::", "").replace("void ", "")
    return n


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    disp = OrderedDict()
    for r in rows:
        d = disp.setdefault(r["Dispatch_Id"], {"name": short(r["Kernel_Name"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    agg = OrderedDict()
    for d in disp.values():
        a = agg.setdefault(d["name"], [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        a[2] += d.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
        a[3] += d["ns"]
    out = ["| kernel | launches | MFMA busy cycles (sum over SIMDs) | GRBM_GUI_ACTIVE (per XCD, sum over launches) "
           "| GUI_ACTIVE / wall ns | MfmaUtil % | MFMA busy % of kernel wall time @2.4 GHz |",
           "|---|---:|---:|---:|---:|---:|---:|"]
    tb = ta = tn = 0.0
    for k, (n, busy, act, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if busy == 0:
            continue
        tb += busy
        ta += act
        tn += ns
        out.append(f"| `{k}` | {n} | {busy:.3e} | {act:.3e} | {act / ns:.2f} | {100 * busy / (act * SIMD_NUM):.1f} "
                   f"| {100 * busy / (ns * 2.4 * SIMD_NUM):.1f} |")
    out.append(f"| **all MFMA kernels** | | {tb:.3e} | {ta:.3e} | {ta / tn:.2f} | {100 * tb / (ta * SIMD_NUM):.1f} "
               f"| {100 * tb / (tn * 2.4 * SIMD_NUM):.1f} |")
    out.append("")
    out.append("GUI_ACTIVE/wall > 2.4 'GHz' shows GRBM_GUI_ACTIVE also covers the serialised PMC dispatch's idle lead-in/out, "
               "so MfmaUtil (rocprofv3's formula) under-reads short kernels; the last column divides by the kernel's own "
               "End-Start time at the 2.4 GHz peak clock instead (a lower bound when the chip clocks below peak).")
    txt = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
