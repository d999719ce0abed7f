#!/usr/bin/env python
"""Small driver for the rocprofv3 PMC passes: 3 TemporalUNet forwards at B=1024 (the conv kernel family is 95% of
the hot path's GPU time; the full bench.py crashes rocprofv3's counter mode on this image)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from edmp_amd import _capi
from edmp_amd.runtime import ptr
from edmp_amd.temporalunet import TemporalUNet

B = int(os.environ.get("EDMP_PMC_BATCH", "1024"))
net = TemporalUNet(None, 7, 32, "cuda:0", dims=(32, 64, 128, 256, 512, 512), seed=1, max_batch=B)
ctx = net.ctx
x = ctx.to_dev(torch.randn(B, 7, 50), torch.float32)
eps = ctx.empty(x.shape, torch.float32)
for _ in range(int(os.environ.get("EDMP_PMC_FORWARDS", "3"))):
    _capi.check(ctx.lib.edmp_unet_forward_dev(ctx.h, ptr(x), B, 100, ptr(eps)))
ctx.sync()
print("done")
