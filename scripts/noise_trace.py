#!/usr/bin/env python
"""timeline of one denoise_guided(noise=None) call: per chunk, when the draw started / ended, when the main thread got it,
uploaded it, finished enqueueing its launches, and when the GPU finished it (ms since the call started)."""
import os
import sys
import time

os.environ["EDMP_NOISE_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from edmp_amd import guide_cfg as GC, scenes  # noqa: E402
from edmp_amd.diffusion import Diffusion  # noqa: E402
from edmp_amd.guide import IntersectionVolumeGuide  # noqa: E402
from edmp_amd.temporalunet import TemporalUNet  # noqa: E402

T, N, C, B = 255, 50, 7, 1024
dev = "cuda:0"
guides = [1, 2, 3, 4, 5, 10]
cfgs = GC.build_guide_cfgs([GC.catalog_guide_dict(g) for g in guides], 0, T, rows_per_guide=GC.split_rows(B, len(guides)))
net = TemporalUNet(None, C, 32, dev, dims=(32, 64, 128, 256, 512, 512), seed=1, max_batch=B)
guide = IntersectionVolumeGuide(scenes.random_scene(11, 16), dev, cfgs, B)
dif = Diffusion(T, dev)
start, goal = scenes.DEFAULT_START, scenes.DEFAULT_GOAL
runs = []
for r in range(int(os.environ.get("RUNS", "6"))):
    np.random.seed(0)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record(dif.ctx.stream)
    t0 = time.perf_counter()
    X = dif.denoise_guided(net, guide, N, C, cfgs["guidance_schedule"], batch_size=B, start=start, goal=goal, return_device=True, chunk_steps=int(os.environ.get("CHUNK", "16")))
    torch.cuda.synchronize()
    runs.append((time.perf_counter() - t0, e0, dif.noise_trace))
runs = runs[1:]
for tag, (dt, e0, tr) in (("fastest", min(runs, key=lambda x: x[0])), ("slowest", max(runs, key=lambda x: x[0]))):
    print(f"=== {tag} run: {1e3 * dt:.1f} ms; all runs: {[round(1e3 * r[0]) for r in runs]}")
    draws = {x[1]: x for x in tr if x[0] == "draw"}
    print("chunk steps | draw: wait-slot start end | main: wait got uploaded enqueued | gpu done | gpu idle before chunk?")
    prev_gpu = 0.0
    for x in tr:
        if x[0] != "main":
            continue
        _, i, kk, ta, tb, tc, td, ev = x
        d = draws[i]
        gpu_done = e0.elapsed_time(ev)
        print(f"{i:3d} {kk:3d} | {1e3 * d[3]:7.1f} {1e3 * d[4]:7.1f} {1e3 * d[5]:7.1f} | {1e3 * ta:7.1f} {1e3 * tb:7.1f} {1e3 * tc:7.1f} {1e3 * td:7.1f} | {gpu_done:7.1f} | gpu time {gpu_done - prev_gpu:6.1f} ms for {kk} steps")
        prev_gpu = gpu_done
