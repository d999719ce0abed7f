#!/usr/bin/env python
"""Phase timing of the fused conv kernels: run UNet forwards against a library built with -DEDMP_STAMPS
(hipcc ... -DEDMP_STAMPS libedmp_hip.hip -o <lib>; EDMP_STAMP_LIB=<lib>) and print, per instrumented kernel, the
s_memtime deltas between the phase boundaries of ONE mid-grid workgroup (prologue | K loop | spill | statistics | output
pass) in shader cycles and in 10 ns wall ticks.  This is how the epilogue stalls fixed in round 1 were found."""
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from edmp_amd import _capi
_capi.LIB_PATH = os.environ.get("EDMP_STAMP_LIB", os.path.join(os.getcwd(), "scratch/stamps/libedmp_hip.so"))  # built with -DEDMP_STAMPS
from edmp_amd.runtime import ptr
from edmp_amd.temporalunet import TemporalUNet
B = 1024
net = TemporalUNet(None, 7, 32, "cuda:0", dims=(32, 64, 128, 256, 512, 512), seed=1, max_batch=B)
ctx = net.ctx
x = ctx.to_dev(torch.randn(B, 7, 50), torch.float32)
eps = ctx.empty(x.shape, torch.float32)
names = {0: "rows<64,13> (last: up? no: down2.rcb1.conv2 Cin128)", 1: "rows<64,7,9>", 2: "wide<64,2>", 3: "wide<64,4>", 4: "wide<32,7>", 5: "wide<32,4>", 6: "conv_mfma<64,64,64> (last launch)", 7: "block<64,25,4,16,id>"}
for rep in range(4):
    _capi.check(ctx.lib.edmp_unet_forward_dev(ctx.h, ptr(x), B, 100, ptr(eps)))
    ctx.sync()
    buf = (C.c_ulonglong * 128)()
    ctx.lib.edmp_debug_stamps.argtypes = [C.c_void_p]
    rc = ctx.lib.edmp_debug_stamps(buf)
    a = np.array(buf[:], dtype=np.uint64).reshape(8, 8, 2).astype(np.int64)
    if rep < 3: continue
    for k in range(8):
        cyc = a[k, :, 0]; wall = a[k, :, 1]
        if cyc[0] == 0: continue
        n = 6 if k in (0, 1) else (8 if k == 7 else (4 if k == 6 else 5))
        dc = np.diff(cyc[:n]); dw = np.diff(wall[:n])
        raw = np.array(buf[:], dtype=np.uint64).reshape(8, 16).astype(np.int64)
        if raw[k, 14] > 0: print(f"    in-loop (tid0): fetch {raw[k,10]} compute {raw[k,11]} commit {raw[k,12]} barrier {raw[k,13]} nK {raw[k,14]}")
        print(f"{names[k]:50s} cycles: {dc.tolist()} total {cyc[n-1]-cyc[0]}  wall(10ns): {dw.tolist()} total {wall[n-1]-wall[0]}")
