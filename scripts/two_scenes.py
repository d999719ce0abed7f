#!/usr/bin/env python
"""Experiment: TWO independent 1024-row scenes in flight on one GPU (two contexts = two streams, two host threads) against one
scene at a time.  Every launch of the sampler is one wave of 256 workgroups, so a single chain leaves the chip idle in every
dispatch gap and under-filled in every kernel tail; a second independent chain can fill those holes."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from edmp_amd import guide_cfg as GC, scenes  # noqa: E402
from edmp_amd.diffusion import Diffusion  # noqa: E402
from edmp_amd.guide import IntersectionVolumeGuide  # noqa: E402
from edmp_amd.runtime import Context, get_context  # noqa: E402
from edmp_amd.temporalunet import TemporalUNet  # noqa: E402

T, N, C, B = 255, 50, 7, 1024
guides = [1, 2, 3, 4, 5, 10]
cfgs = GC.build_guide_cfgs([GC.catalog_guide_dict(g) for g in guides], 0, T, rows_per_guide=GC.split_rows(B, len(guides)))
start, goal = scenes.DEFAULT_START, scenes.DEFAULT_GOAL


def make(ctx, seed):
    net = TemporalUNet(None, C, 32, ctx, dims=(32, 64, 128, 256, 512, 512), seed=1, max_batch=B)
    guide = IntersectionVolumeGuide(scenes.random_scene(11 + seed, 16), ctx, cfgs, B)
    dif = Diffusion(T, ctx)
    noise = dif.ctx.to_dev(np.random.RandomState(seed).standard_normal((T + 1, B, C, N)), torch.float64)
    dif.ctx.sync()

    def run(k):
        for _ in range(k):
            X = dif.denoise_guided(net, guide, N, C, cfgs["guidance_schedule"], batch_size=B, start=start, goal=goal, noise=noise, return_device=True)
            guide.row_swept_volumes(start, goal, X)
            guide.success_rows(X)
        return X

    return run


a = make(get_context("cuda:0"), 1)
b = make(Context(0), 2)
a(1), b(1)
torch.cuda.synchronize()
K = 3
t0 = time.perf_counter()
Xa = a(K)
torch.cuda.synchronize()
t1 = time.perf_counter() - t0
t0 = time.perf_counter()
ths = [threading.Thread(target=f, args=(K,)) for f in (a, b)]
for th in ths:
    th.start()
for th in ths:
    th.join()
torch.cuda.synchronize()
t2 = time.perf_counter() - t0
print(f"one scene at a time: {1e3 * t1 / K:.1f} ms per scene = {B * T * K / t1 / 1e3:.0f} k traj-steps/s")
print(f"two scenes in flight: {1e3 * t2 / K:.1f} ms per pair = {2 * B * T * K / t2 / 1e3:.0f} k traj-steps/s  (x{2 * t1 / t2:.3f})")
Xa2 = a(1)
print("scene A bit-identical alone vs concurrent:", bool(torch.equal(Xa, Xa2)))
