#!/usr/bin/env python
"""A/B of builder switches in ONE process on ONE box: every variant is a model built under its own environment (the
switches are frozen into the layer program at build time), the variants are timed alternately.

    python scripts/ab_models.py base: sb2:EDMP_LEVEL_SB=2222 nokar:EDMP_NO_KARATSUBA=1

Prints per variant the steady-state forward time (HIP events around 20 forwards, best of the rounds) and the per-kernel
table of the whole-level / position-tile kernels (HIP-event brackets per launch, edmp_prof_ops)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from edmp_amd.temporalunet import TemporalUNet  # noqa: E402

FULL = (32, 64, 128, 256, 512, 512)
B = int(os.environ.get("AB_BATCH", "1024"))
specs = []
for a in sys.argv[1:]:
    name, _, envs = a.partition(":")
    specs.append((name, dict(e.split("=", 1) for e in envs.split(",") if e)))
models = []
for name, env in specs:
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    models.append((name, TemporalUNet(None, 7, 32, "cuda:0", dims=FULL, seed=1, max_batch=B)))
    for k, v in old.items():
        if v is None:
            os.environ.pop(k)
        else:
            os.environ[k] = v
x = torch.randn(B, 7, 50, device="cuda:0")
t = torch.tensor([100.0])
ref = None
best = {n: 1e9 for n, _ in models}
for rnd in range(4):
    for name, m in models:
        for _ in range(3):
            y = m(x, t)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            y = m(x, t)
        e1.record()
        torch.cuda.synchronize()
        best[name] = min(best[name], e0.elapsed_time(e1) / 20)
        if ref is None:
            ref = y.cpu().numpy()
        else:
            d = float(np.sqrt(np.mean((y.cpu().numpy() - ref) ** 2)))
            assert d < 1e-5, (name, d)
for name, m in models:
    print(f"{name:12s} forward {1e3 * best[name]:8.1f} us")
for name, m in models:
    ctx = m.ctx
    m(x, t)
    ctx.prof(1)
    ctx.prof_read(reset=True)
    for _ in range(10):
        m(x, t)
    ops = ctx.prof_ops()
    ctx.prof_read(reset=True)
    ctx.prof(0)
    tab = {}
    for nm, calls, ms, fl in ops:
        if calls:
            r = tab.setdefault(nm, [0, 0.0])
            r[0] += calls
            r[1] += ms
    tot = sum(v[1] for v in tab.values()) / 10
    print(f"--- {name}: sum of per-launch brackets {1e3 * tot:.1f} us per forward")
    for nm, (c, ms) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
        if "level" in nm or os.environ.get("AB_ALL"):
            print(f"   {nm:44s} n/fwd {c // 10:3d}  avg {1e3 * ms / c:7.2f} us")
