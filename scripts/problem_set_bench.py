#!/usr/bin/env python
"""problem_set_bench.py — throughput over a PROBLEM SET, counted the way the reference counts a scene.

`BASELINE.json:metric` is quoted "on the global_solvable_problems set"; the reference's per-scene clock ("Planning Time",
infer_serial.py:108-157) covers guide construction + the IK-goal filter + sampling + the best-trajectory pick.  `bench.py` builds
the guide once outside its timed loop; this script runs `infer_serial.run` - the reference-shaped scene loop - over >= 16 DISTINCT
synthetic scenes at BASELINE config 3's size (1024 rows, guides [1,2,3,4,5,10], 16 obstacles of which 3 true cylinders, 100 IK
candidates per scene, noise drawn per scene from NumPy's global RandomState exactly as the reference does), serially and with two
scenes in flight, and reports scenes/s, traj-steps/s and where a scene's time goes.

    python scripts/problem_set_bench.py [--scenes 16] [--out profiles/r05_problem_set.json]

Prints ONE JSON object (also written to --out).  Informative: never bench.py's `value`."""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import yaml  # noqa: E402

T, N, C = 255, 50, 7


def measure(n_scenes=16, rows=1024, guides=(1, 2, 3, 4, 5, 10), n_obstacles=16, n_cylinders=3, flights=(1, 2), device="cuda:0", seed=0, verbose=False):
    import torch

    import infer_serial
    from edmp_amd.scenes import SyntheticDataset

    cfg = {
        "guide": {"guides": list(guides), "batch_size_per_guide": rows // len(guides), "total_rows": rows, "guide_path": "./guides/"},
        "dataset": {"path": "./datasets/", "dataset_type": "synthetic", "scene_types": ["tabletop", "stress"], "num_scenes_per_type": (n_scenes + 1) // 2},
        "model": {"model_dir": "./models/", "device": device, "T": T, "traj_len": N, "num_channels": C},
        "general": {"gui": False},
    }
    out = {"scenes": n_scenes, "rows_per_scene": rows, "guides": list(guides), "obstacles": n_obstacles, "true_cylinders": n_cylinders,
           "noise": "NumPy global RandomState per scene (the reference's contract), drawn by edmp_amd.nprng bit for bit",
           "clock": "wall time of infer_serial.run's scene loop (the run's model build / upload excluded) over all scenes; steady_state_* = between the end of the first group of "
                    "k scenes in flight and the end of the last (n - k scenes); per scene the reference's Planning Time split"}
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "configs"))
        path = os.path.join(td, "configs", "cfg_problem_set.yaml")
        with open(path, "w") as f:
            yaml.safe_dump(cfg, f)
        ds = SyntheticDataset("synthetic", scene_types=("tabletop", "stress"), num_scenes_per_type=(n_scenes + 1) // 2, n_obstacles=n_obstacles, n_cylinders=n_cylinders)
        # one untimed scene per lane count: model upload, draw-thread team, kernel attributes
        for k in flights:
            np.random.seed(seed)
            infer_serial.run(path, dataset=ds, max_scenes=k, verbose=False, scenes_in_flight=k, shard_scenes=False)
        for k in flights:
            np.random.seed(seed)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = infer_serial.run(path, dataset=ds, max_scenes=n_scenes, verbose=verbose, scenes_in_flight=k, shard_scenes=False)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0 - infer_serial.run.last_setup_s  # (config parse, model build + upload: per run, not per scene)
            assert len(res) == n_scenes
            # steady state: scenes in flight finish in groups of k (they share the GPU from start to end), so the rate is taken between
            # the END of the first group and the END of the last: n - k scenes in that interval (an interval that starts at the first
            # completion of a group and ends at the last completion of another counts one group too many: +8 % at k = 2, n = 16)
            done = [r["done_at"] for r in res]
            steady = (n_scenes - k) / (done[-1] - done[k - 1]) if n_scenes > k else n_scenes / wall
            keys = sorted(res[0]["timings"])
            split = {key: float(np.mean([r["timings"][key] for r in res])) for key in keys}
            plan = float(np.mean([r["planning_time_s"] for r in res]))
            out["serial" if k == 1 else f"scenes_in_flight_{k}"] = {
                "wall_s": wall, "scenes_per_s": n_scenes / wall, "traj_steps_per_s": n_scenes * rows * T / wall,
                "steady_state_scenes_per_s": steady, "steady_state_traj_steps_per_s": steady * rows * T,
                "planning_time_s_mean": plan, "scene_wall_s_mean": float(np.mean([r["scene_wall_s"] for r in res])),
                "per_scene_split_s_mean": split,
                "share_of_planning_time": {key: split[key] / plan for key in keys},
                "success_proxy_collision_free": int(sum(r["success_proxy"] for r in res)),
            }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=16)
    ap.add_argument("--rows", type=int, default=1024)
    ap.add_argument("--out", type=str, default=None)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--flights", type=str, default="1,2", help="scenes in flight to measure, e.g. 1,2,3")
    a = ap.parse_args()
    out = measure(a.scenes, a.rows, flights=tuple(int(k) for k in a.flights.split(",")), verbose=a.verbose)
    txt = json.dumps(out, indent=1)
    print(txt)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
