#!/usr/bin/env python
"""The numbers profiles/README.md and DESIGN.md quote for a round, DERIVED from the committed files profiles/<round>_bench.json,
<round>_kernel_stats.csv, <round>_pmc_hbm_traffic.json and <round>_pmc_mfma_util.md (no hand-typed figures).
Usage: python scripts/round_numbers.py r04 > profiles/r04_numbers.md"""
import csv
import json
import os
import re
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
b = json.load(open(os.path.join(P, f"{R}_bench.json")))
rf = b["roofline"]
print(f"# {R} - numbers derived from the committed profile files (scripts/round_numbers.py {R})\n")
print("## bench.py line (`%s_bench.json`)\n" % R)
print(f"* value **{b['value']:.0f} {b['unit']}**, ms_per_step {b['ms_per_step']:.2f}, n_gpus {b['n_gpus']}")
print(f"* roofline: issued {rf['achieved']:.1f} TFLOP/s = **{rf['frac']:.3f}** of {rf['peak']}; direct form {rf['achieved_direct_form']:.1f} = {rf['frac_direct_form']:.3f}; nominal {rf['achieved_nominal']:.1f}; "
      f"conv {rf['conv_ms_per_call']:.1f} ms per call over {rf['launches']} launches (avg {rf['avg_launch_us']:.2f} us), share of the step {rf['conv_share_of_step']:.3f}")
if rf.get("hbm"):
    print(f"* HBM: {rf['traffic'] / 1e6:.1f} MB per conv launch ({rf['traffic_detail']['source']}), {rf['hbm']['achieved']:.0f} GB/s = {rf['hbm']['frac']:.3f} of 8 TB/s; "
          f"{rf['traffic_detail']['hbm_bytes_per_traj_step'] / 1e6:.2f} MB per trajectory-step")
sp = b["success_proxy"]
print(f"* success proxy: collision-free {sp['rows_collision_free']}/{sp['rows']} (the reference's criterion), strict {sp['rows_ok']}/{sp['rows']}; check kernel "
      f"{sp.get('check_ms', {}).get('this_batch', float('nan')):.3f} ms on this batch, {sp.get('check_ms', {}).get('collision_free_batch_worst_case', float('nan')):.3f} ms on a collision-free batch")
e = b.get("end_to_end_scene_seconds")
if e:
    print(f"* scene seconds: noise resident {e['noise_resident_in_hbm']:.4f}, NumPy stream drawn + uploaded {e['numpy_stream_drawn_and_uploaded_per_scene']:.4f} "
          f"(first scene of the process {e['numpy_stream_first_scene_of_the_process']:.4f}), device Philox {e['device_philox_noise']:.4f}")
t = b.get("two_scenes_in_flight")
if t:
    print(f"* two scenes in flight: {t['traj_steps_per_s']:.0f} traj-steps/s = x{t['vs_value']:.3f}")
ps = b.get("problem_set")
if ps:
    for key in ("serial", "scenes_in_flight_2"):
        q = ps.get(key)
        if q:
            split = ", ".join(f"{k[:-2]} {1e3 * v:.1f} ms" for k, v in q["per_scene_split_s_mean"].items())
            print(f"* problem set ({ps['scenes']} scenes x {ps['rows_per_scene']} rows, {key}): {q['scenes_per_s']:.2f} scenes/s = {q['traj_steps_per_s']:.0f} traj-steps/s (steady state {q.get('steady_state_traj_steps_per_s', float('nan')):.0f}); "
                  f"planning time {1e3 * q['planning_time_s_mean']:.1f} ms per scene: {split}")
for tag in ("c2", "c5_n1", "c5_n1_pyhook"):
    fp = os.path.join(P, f"{R}_bench_{tag}.json")
    if os.path.exists(fp):
        q = json.load(open(fp))
        print(f"* `{R}_bench_{tag}.json`: {q['value']:.0f} {q['unit']}, ms_per_step {q['ms_per_step']:.2f}, roofline frac {q.get('roofline', {}).get('frac', float('nan')):.3f} - {q['config']['workload'][:110]}"
              + (f"; all-reduce hook ({q['allreduce_hook'].get('kind', 'python callback')}) {q['allreduce_hook']['avg_us']:.1f} us avg x {q['allreduce_hook']['calls_per_denoise']} calls" if q.get("allreduce_hook") else ""))
c = b.get("cpu_baseline")
if c:
    print(f"* cpu_baseline ({c['kind']}): {c['value']:.0f} {c['unit']} on {c['cores']} cores; {c['sample']}")
print("\n| kernel | launches | avg us | share | TFLOP/s issued | frac |\n|---|---:|---:|---:|---:|---:|")
for r in rf["per_kernel"]:
    print(f"| `{r['kernel']}` | {r['launches']} | {r['avg_us']:.2f} | {r['share']:.3f} | {r.get('executed_tflops', r.get('issued_tflops', 0.0)):.1f}{' (bf16 pipe)' if r.get('pipe') == 'bf16' else ''} | {r['frac']:.3f} |")
lv = [r for r in rf["per_kernel"] if r["kernel"].startswith(("level_kernel", "level2_kernel"))]
print(f"\nLevel kernels per reverse step: {sum(r['avg_us'] for r in lv):.1f} us in {len(lv)} launches.")

print(f"\n## rocprofv3 kernel trace of the same command (`{R}_kernel_stats.csv`)\n")
rows = list(csv.DictReader(open(os.path.join(P, f"{R}_kernel_stats.csv"))))
conv = [r for r in rows if "wide_conv_kernel" in r["Name"] or "level_kernel" in r["Name"] or "level2_kernel" in r["Name"]]
tot_ns = sum(int(r["TotalDurationNs"]) for r in conv)
calls = sum(int(r["Calls"]) for r in conv)
dom = [r for r in rows if "wide_conv_kernel<3, 32, 64, 64, 2, false>" in r["Name"].replace("(edmp::WideKind)", "")]
per_call_launches = rf["launches"]
ncalls = calls / per_call_launches
print(f"* conv family: {tot_ns / 1e6:.2f} ms in {calls} launches = {ncalls:.2f} `denoise_guided` calls -> {tot_ns / 1e6 / ncalls:.2f} ms per call, {tot_ns / 1e3 / calls:.2f} us average "
      f"(bench live brackets: {rf['conv_ms_per_call']:.2f} ms, {rf['avg_launch_us']:.2f} us)")
for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"]))[:6]:
    nm = re.sub(r"\(edmp::WideKind\)|\(edmp::LevelMode\)|void |edmp::", "", r["Name"].split("(")[0] if "<" not in r["Name"] else r["Name"])
    print(f"* `{nm[:70]}`: {r['Calls']} calls, avg {float(r['AverageNs']) / 1e3:.2f} us, {float(r['Percentage']):.1f} %")

print(f"\n## PMC passes (`{R}_pmc_hbm_traffic.json`, `{R}_pmc_mfma_util.md`)\n")
h = json.load(open(os.path.join(P, f"{R}_pmc_hbm_traffic.json")))["conv_family"]
print(f"* HBM traffic: {h['hbm_MB_per_forward']:.0f} MB per forward over {h['launches_per_forward']} launches = {h['hbm_MB_per_launch']:.1f} MB per launch, {h['hbm_bytes_per_traj_step'] / 1e6:.3f} MB per trajectory-step "
      f"(algorithmic 0.121 MB: x{h['hbm_bytes_per_traj_step'] / 121147:.1f})")
for line in open(os.path.join(P, f"{R}_pmc_mfma_util.md")):
    if line.startswith(("| `level_kernel", "| `level2_kernel")) or line.startswith("| **all") or "<3, 32, 64, 64, 2, false>" in line:
        c = [x.strip() for x in line.strip().strip("|").split("|")]
        print(f"* MFMA busy of kernel wall time @2.4 GHz: {c[0]} -> {c[-1]} %")
