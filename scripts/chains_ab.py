#!/usr/bin/env python
"""Same-box A/B of row chains of ONE 1024-row batch: denoise_guided(chains=k) for k in argv (default 1 2 3 4),
alternated, best of the rounds; also two independent scenes in flight on two contexts for comparison.  Prints traj-steps/s."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from edmp_amd import guide_cfg as GC  # noqa: E402
from edmp_amd import scenes  # noqa: E402
from edmp_amd.diffusion import Diffusion  # noqa: E402
from edmp_amd.guide import IntersectionVolumeGuide  # noqa: E402
from edmp_amd.temporalunet import TemporalUNet  # noqa: E402

T, N, C, B = 255, 50, 7, int(os.environ.get("AB_BATCH", "1024"))
ks = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
guides = [1, 2, 3, 4, 5, 10]
cfgs = GC.build_guide_cfgs([GC.catalog_guide_dict(g) for g in guides], 0, T, rows_per_guide=GC.split_rows(B, len(guides)))
net = TemporalUNet(None, C, 32, "cuda:0", dims=(32, 64, 128, 256, 512, 512), seed=1, max_batch=B)
guide = IntersectionVolumeGuide(scenes.random_scene(11, 16), "cuda:0", cfgs, B)
dif = Diffusion(T, "cuda:0")
noise = dif.ctx.to_dev(np.random.RandomState(1234).standard_normal((T + 1, B, C, N)), torch.float64)
dif.ctx.sync()


def run(k):
    t0 = time.perf_counter()
    X = dif.denoise_guided(net, guide, N, C, cfgs["guidance_schedule"], batch_size=B, start=scenes.DEFAULT_START, goal=scenes.DEFAULT_GOAL, noise=noise, return_device=True, chains=k)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, X


ref = run(1)[1].cpu().numpy()
best = {k: 1e9 for k in ks}
for rnd in range(int(os.environ.get("AB_ROUNDS", "3"))):
    for k in ks:
        dt, X = run(k)
        best[k] = min(best[k], dt)
        if rnd == 0:
            assert np.array_equal(X.cpu().numpy(), ref), k
for k in ks:
    print(f"chains={k}: {1e3 * best[k]:8.2f} ms per scene  {B * T / best[k]:10.0f} traj-steps/s  x{best[ks[0]] / best[k]:.4f} (bit-identical to chains=1)")
