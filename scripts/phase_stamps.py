#!/usr/bin/env python
"""Phase timing of the conv kernels: run UNet forwards against a library built with -DEDMP_STAMPS
(hipcc ... -DEDMP_STAMPS libedmp_hip.hip -o scratch/stamps/libedmp_hip.so; EDMP_STAMP_LIB=<lib>) and print, per instrumented
kernel, the s_memtime deltas between the phase boundaries of workgroup 0 in shader cycles and the total in microseconds.
Slot 0: wide_conv_kernel (last launch of the forward); slots 1-4: the four level_kernel variants."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from edmp_amd import _capi

_capi.LIB_PATH = os.environ.get("EDMP_STAMP_LIB", os.path.join(os.getcwd(), "scratch/stamps/libedmp_hip.so"))
from edmp_amd.runtime import ptr
from edmp_amd.temporalunet import TemporalUNet

B = 1024
net = TemporalUNet(None, 7, 32, "cuda:0", dims=(32, 64, 128, 256, 512, 512), seed=1, max_batch=B)
ctx = net.ctx
x = ctx.to_dev(torch.randn(B, 7, 50), torch.float32)
eps = ctx.empty(x.shape, torch.float32)
names = {0: "wide_conv (last launch)", 1: "level DOWN 32@50", 2: "level DOWN 64@25", 3: "level UP 64@13", 4: "level UP_FINAL 32@25"}
labels = {0: ["prologue", "K loop", "spill", "stats+store"],
          1: ["zero+load", "conv11", "epi11", "conv12", "epi12", "RCB2", "resample(+final)"]}
for rep in range(4):
    _capi.check(ctx.lib.edmp_unet_forward_dev(ctx.h, ptr(x), B, 100, ptr(eps)))
    ctx.sync()
buf = (C.c_ulonglong * 128)()
ctx.lib.edmp_debug_stamps.argtypes = [C.c_void_p]
ctx.lib.edmp_debug_stamps(buf)
a = np.array(buf[:], dtype=np.uint64).reshape(8, 8, 2).astype(np.int64)
for k in range(5):
    cyc, wall = a[k, :, 0], a[k, :, 1]
    if cyc[0] == 0:
        continue
    n = 5 if k == 0 else 8
    dc = np.diff(cyc[:n])
    lab = labels[0] if k == 0 else labels[1]
    print(f"{names[k]:24s} " + " | ".join(f"{l} {int(v)}" for l, v in zip(lab, dc)) + f" | total {int(cyc[n - 1] - cyc[0])} cyc = {(wall[n - 1] - wall[0]) / 100.0:.2f} us")
