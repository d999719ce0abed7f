"""shared helpers for the parity tests (test infrastructure)."""
import json

import numpy as np

from edmp_amd import guide_cfg as GC

T = 255
TINY_DIMS = (16, 16, 32, 32, 64, 64)
FULL_DIMS = (32, 64, 128, 256, 512, 512)


def cfgs_for(guides, bpg, rows_per_guide=None):
    return GC.build_guide_cfgs([GC.catalog_guide_dict(int(n)) for n in guides], int(bpg), T, rows_per_guide)


def rmse(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)))


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


def noise_for(seed, B):
    """the reference's RNG stream for one denoise_guided call (np.random.seed(seed) first)."""
    np.random.seed(int(seed))
    return np.random.standard_normal((T + 1, B, 7, 50))
