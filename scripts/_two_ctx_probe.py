import sys, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from edmp_amd import guide_cfg as GC, scenes
from edmp_amd.diffusion import Diffusion
from edmp_amd.guide import IntersectionVolumeGuide
from edmp_amd.temporalunet import TemporalUNet
from edmp_amd.runtime import lane_context, get_context
T, B = 255, int(os.environ.get("PB", "1024"))
FULL = (32, 64, 128, 256, 512, 512)
guides = [1, 2, 3, 4, 5, 10]
cfgs = GC.build_guide_cfgs([GC.catalog_guide_dict(g) for g in guides], 0, T, rows_per_guide=GC.split_rows(B, len(guides)))
steps = int(os.environ.get("STEPS", "2"))
objs = []
for lane in (0, 1):
    ctx = get_context("cuda:0") if lane == 0 else lane_context(0, 1)
    net = TemporalUNet(None, 7, 32, ctx, dims=FULL, seed=1, max_batch=B)
    guide = IntersectionVolumeGuide(scenes.random_scene(11 + lane, 16), ctx, cfgs, B)
    dif = Diffusion(T, ctx)
    noise = ctx.to_dev(np.random.RandomState(99 + lane).standard_normal((T + 1, B, 7, 50)), torch.float64)
    kw = dict(batch_size=B, start=scenes.DEFAULT_START, goal=scenes.DEFAULT_GOAL, noise=noise, t_stop=T - steps)
    ref = dif.denoise_guided(net, guide, 50, 7, cfgs["guidance_schedule"], **kw)
    objs.append((dif, net, guide, kw, ref))
bad = [0, 0]
REPS = int(os.environ.get("REPS", "60"))
def work(lane):
    dif, net, guide, kw, ref = objs[lane]
    for rep in range(REPS):
        X = dif.denoise_guided(net, guide, 50, 7, cfgs["guidance_schedule"], **kw)
        if not np.array_equal(X, ref):
            bad[lane] += 1
            if bad[lane] <= 3:
                rows = np.unique(np.nonzero(X != ref)[0])
                print(f"  lane {lane} rep {rep}: rows {rows[:12]} max {np.abs(X - ref).max():.3e}", flush=True)
ths = [threading.Thread(target=work, args=(l,)) for l in (0, 1)]
[t.start() for t in ths]; [t.join() for t in ths]
print(f"two contexts concurrently, {steps} steps: runs that differ from the serial reference: {bad} of {REPS} each", flush=True)
