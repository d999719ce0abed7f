#!/usr/bin/env python
"""Where the per-scene host work of the reference-shaped loop goes (infer_serial.py:108-129: guide object + IK filter): cProfile over
N scene changes at BASELINE config 3's size.  python scripts/guide_ctor_profile.py [n = 20]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from edmp_amd import guide_cfg as GC  # noqa: E402
from edmp_amd.guide import IntersectionVolumeGuide  # noqa: E402
from edmp_amd.scenes import SyntheticDataset  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfgs = GC.build_guide_cfgs([GC.catalog_guide_dict(g) for g in (1, 2, 3, 4, 5, 10)], 0, 255, rows_per_guide=GC.split_rows(1024, 6))
ds = SyntheticDataset("synthetic", scene_types=("stress",), num_scenes_per_type=n + 2, n_obstacles=16, n_cylinders=3)


def scene(i):
    oc, _, _, nb, nc, start, iks = ds.fetch_data(i, "stress")
    kinds = np.concatenate([np.zeros(nb, dtype=np.int32), np.ones(nc, dtype=np.int32)])
    t0 = time.perf_counter()
    g = IntersectionVolumeGuide(oc, "cuda:0", cfgs, 1024, obstacle_kinds=kinds)
    t1 = time.perf_counter()
    v = g.cost(torch.tensor(iks.reshape((-1, 7, 1))), 0, batch_size=iks.shape[0]).sum(axis=(1, 2)).cpu().numpy()
    return t1 - t0, time.perf_counter() - t1, v


scene(0), scene(1)
pr = cProfile.Profile()
pr.enable()
tt = [scene(2 + i)[:2] for i in range(n)]
pr.disable()
print(f"guide ctor {1e3 * np.mean([a for a, _ in tt]):.3f} ms, IK filter cost {1e3 * np.mean([b for _, b in tt]):.3f} ms per scene (mean of {n})")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
