#!/usr/bin/env python
"""Same-box A/B of whole-run hipGraph replay (edmp_sampler_set_graph) on the bench workload: B = 1024, six guides, noise resident."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from edmp_amd import guide_cfg as GC, scenes
from edmp_amd.diffusion import Diffusion
from edmp_amd.guide import IntersectionVolumeGuide
from edmp_amd.temporalunet import TemporalUNet

T, N, C, B, dev = 255, 50, 7, 1024, "cuda:0"
guides = [1, 2, 3, 4, 5, 10]
cfgs = GC.build_guide_cfgs([GC.catalog_guide_dict(g) for g in guides], 0, T, rows_per_guide=GC.split_rows(B, len(guides)))
net = TemporalUNet(None, C, 32, dev, dims=(32, 64, 128, 256, 512, 512), seed=1, max_batch=B)
guide = IntersectionVolumeGuide(scenes.random_scene(11, 16), dev, cfgs, B)
dif = Diffusion(T, dev)
ctx = dif.ctx
noise = ctx.to_dev(np.random.RandomState(1234).standard_normal((T + 1, B, C, N)), torch.float64)
kw = dict(batch_size=B, start=scenes.DEFAULT_START, goal=scenes.DEFAULT_GOAL, noise=noise, return_device=True)
def run(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k):
        X = dif.denoise_guided(net, guide, N, C, cfgs["guidance_schedule"], **kw)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k, X
ref = None
for rnd in range(3):
    for on in (0, 1):
        ctx.lib.edmp_sampler_set_graph(ctx.h, on)
        run(2)  # (graph: capture + first replay)
        dt, X = run(3)
        if ref is None: ref = X.clone()
        print(f"graph={on}: {1e3 * dt:8.2f} ms per call  {B * T / dt:10.0f} traj-steps/s  identical={bool(torch.equal(X, ref))}", flush=True)
