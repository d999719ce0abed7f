#!/usr/bin/env python
"""Where a scene's time goes under the reference's own noise contract (NumPy global RandomState, diffusion.py:126,303):
raw draw rate of edmp_amd.nprng per thread count, then whole-scene wall time of denoise_guided(noise=None) against the
noise-resident call, for several draw-thread counts / chunk sizes.  Run on the GPU box."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from edmp_amd import guide_cfg as GC, nprng, scenes  # noqa: E402
from edmp_amd.diffusion import Diffusion  # noqa: E402
from edmp_amd.guide import IntersectionVolumeGuide  # noqa: E402
from edmp_amd.temporalunet import TemporalUNet  # noqa: E402

T, N, C, B = 255, 50, 7, 1024
print("cpu quota threads:", nprng.threads(), " affinity:", len(os.sched_getaffinity(0)), " team (threads, domain cores):", nprng.team(nprng.draw_threads()))
np.random.seed(0)
for nt in (1, 4, 8, 12, 14, 16):
    if nt > nprng.threads():
        continue
    n = 358400 * 16
    nprng.standard_normal((n,), nthreads=nt)
    t0 = time.perf_counter()
    nprng.standard_normal((n,), nthreads=nt)
    dt = time.perf_counter() - t0
    print(f"nprng {nt:2d} threads: {1e9 * dt / n:6.2f} ns/normal = {1e3 * dt / 16:6.3f} ms per 1024-row step")
dev = "cuda:0"
guides = [1, 2, 3, 4, 5, 10]
cfgs = GC.build_guide_cfgs([GC.catalog_guide_dict(g) for g in guides], 0, T, rows_per_guide=GC.split_rows(B, len(guides)))
net = TemporalUNet(None, C, 32, dev, dims=(32, 64, 128, 256, 512, 512), seed=1, max_batch=B)
guide = IntersectionVolumeGuide(scenes.random_scene(11, 16), dev, cfgs, B)
dif = Diffusion(T, dev)
start, goal = scenes.DEFAULT_START, scenes.DEFAULT_GOAL
noise = dif.ctx.to_dev(np.random.RandomState(1).standard_normal((T + 1, B, C, N)), torch.float64)


def scene(**kw):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    X = dif.denoise_guided(net, guide, N, C, cfgs["guidance_schedule"], batch_size=B, start=start, goal=goal, return_device=True, **kw)
    guide.row_swept_volumes(start, goal, X)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


scene(noise=noise)
res = min(scene(noise=noise) for _ in range(3))
print(f"noise resident: {1e3 * res:.1f} ms")
for thr in os.environ.get("THREADS", "default,4,6,8").split(","):
    for ck in os.environ.get("CHUNKS", "16").split(","):
        if thr == "default":
            os.environ.pop("EDMP_NPRNG_THREADS", None)
        else:
            os.environ["EDMP_NPRNG_THREADS"] = thr
        np.random.seed(0)
        scene(chunk_steps=int(ck))
        ts = []
        for _ in range(3):
            np.random.seed(0)
            ts.append(scene(chunk_steps=int(ck)))
        print(f"numpy stream, draw threads {thr:>7s}, chunk {ck:>2s}: {1e3 * min(ts):.1f} ms (runs {[round(1e3 * t) for t in ts]}) = +{100 * (min(ts) / res - 1):.1f} % over resident")
