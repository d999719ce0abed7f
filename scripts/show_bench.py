#!/usr/bin/env python
"""Pretty-print a bench.py JSON line (file argument or stdin): headline, roofline summary, per-kernel table."""
import json
import sys

txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
lines = [l for l in txt.strip().splitlines() if l.startswith("{")]
if not lines:
    print("no JSON line found; tail of input:\n" + txt[-600:])
    sys.exit(1)
d = json.loads(lines[-1])
print(f"value {d['value']:.0f} {d['unit']}  ms_per_step {d['ms_per_step']:.2f}  n_gpus {d['n_gpus']}  | {d['config']['parallelism'][:100]}")
if d.get("value_native_f32"):
    print(f"value_native_f32 {d['value_native_f32']:.0f} (all convs on the fp32-MFMA kernels, {d.get('native_f32_leg')}); value_bf16x3 {d.get('value_bf16x3')}; x{d['value'] / d['value_native_f32']:.3f}")
r = d.get("roofline")
if r:
    print(f"roofline: issued {r['achieved']:.1f} TF/s = {r['frac']:.3f} of {r['peak']}; direct-form {r.get('achieved_direct_form', 0):.1f} = {r.get('frac_direct_form', 0):.3f}; nominal {r['achieved_nominal']:.1f}; conv {r['conv_ms_per_call']:.1f} ms "
          f"({r['conv_share_of_step']:.3f} of the step) over {r['launches']} launches, avg {r['avg_launch_us']:.2f} us; timing {r['timing']}")
    for row in r["per_kernel"]:
        tf = row.get("executed_tflops", row.get("issued_tflops", 0.0))
        extra = f" (bf16 pipe; fp32-equivalent {row['fp32_equivalent_tflops']:.1f} TF = {row['fp32_equivalent_vs_fp32_peak']:.2f} of the fp32 peak)" if row.get("pipe") == "bf16" else ""
        print(f"  {row['kernel']:46s} n={row['launches']:5d} avg {row['avg_us']:7.2f} us share {row['share']:.3f} {tf:7.1f} TF frac {row['frac']:.3f}{extra}")
if "cpu_baseline" in d:
    print("cpu_baseline:", d["cpu_baseline"])
if "end_to_end_scene_seconds" in d:
    print("end_to_end:", d["end_to_end_scene_seconds"])
