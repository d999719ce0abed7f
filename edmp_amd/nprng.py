"""NumPy's legacy global RandomState.standard_normal stream, bit for bit, generated in parallel by
libedmp_nprng.so (edmp_amd/csrc/np_legacy_rng.c).  The reference's noise contract is "whatever np.random would have
drawn" (diffusion/diffusion.py:126, 303); NumPy draws the 91.75 M normals of a 1024-trajectory scene one at a time
(0.85 s on the GPU box), which made the host the slower side of the pipeline.  `standard_normal(shape)` reads the global
MT19937 state, produces exactly the values `np.random.standard_normal(shape)` would return, and leaves the global state
exactly where NumPy would leave it, so it can be mixed freely with ordinary np.random calls."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libedmp_nprng.so")
_lib = None


def available() -> bool:
    return os.path.exists(LIB_PATH)


def _load():
    global _lib
    if _lib is None:
        lib = C.CDLL(LIB_PATH)
        lib.edmp_nprng_standard_normal.restype = C.c_int
        lib.edmp_nprng_standard_normal.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double),
                                                   C.POINTER(C.c_double), C.c_int64, C.c_int]
        lib.edmp_nprng_team.restype = C.c_int
        lib.edmp_nprng_team.argtypes = [C.c_int, C.POINTER(C.c_int)]
        _lib = lib
    return _lib


def team(requested: int | None = None):
    """(threads a draw will use, physical cores of the cache domain the team is confined to - 0 if not pinned).  The C
    helper keeps its team inside ONE last-level-cache domain (np_legacy_rng.c: pick_domain) and caps it at that domain's cores."""
    d = C.c_int()
    n = _load().edmp_nprng_team(int(requested or threads()), C.byref(d))
    return int(n), int(d.value)


def threads() -> int:
    """host threads to use: the affinity mask capped by the cgroup CPU quota (a 256-CPU box may grant 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def draw_threads() -> int:
    """threads for the draws made WHILE the GPU loop is being fed: two fewer than the quota, so that the thread that uploads
    and enqueues (and the HIP runtime's own helpers) never push the cgroup over its CPU quota - a throttled cgroup stalls
    every thread for the rest of the scheduler period.  EDMP_NPRNG_THREADS overrides."""
    e = os.environ.get("EDMP_NPRNG_THREADS")
    return max(1, int(e)) if e else max(1, threads() - 2)


def standard_normal(shape, nthreads: int | None = None, out: np.ndarray | None = None) -> np.ndarray:
    """== np.random.standard_normal(shape) on the GLOBAL legacy RandomState (values and state advance), in parallel.
    ``out``: optional C-contiguous float64 array with prod(shape) elements to draw INTO (e.g. a view of pinned host memory
    that is then uploaded by DMA); the returned array is that buffer reshaped.
    Falls back to NumPy itself when the helper library has not been built or the global bit generator is not MT19937."""
    shape = (int(shape),) if np.isscalar(shape) else tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if shape else 1
    if out is not None and (out.dtype != np.float64 or out.size != n or not out.flags["C_CONTIGUOUS"]):
        raise ValueError("out must be a C-contiguous float64 array of prod(shape) elements")
    state = np.random.get_state()
    if n < 4096 or not available() or state[0] != "MT19937":
        v = np.random.standard_normal(shape)
        if out is None:
            return v
        out.reshape(-1)[:] = v.reshape(-1)
        return out.reshape(shape)
    key = np.ascontiguousarray(state[1], dtype=np.uint32).copy()
    pos, has_gauss, gauss = C.c_int(int(state[2])), C.c_int(int(state[3])), C.c_double(float(state[4]))
    out = np.empty(n, dtype=np.float64) if out is None else out.reshape(-1)
    rc = _load().edmp_nprng_standard_normal(key.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos), C.byref(has_gauss), C.byref(gauss),
                                            out.ctypes.data_as(C.POINTER(C.c_double)), n, int(nthreads or threads()))
    if rc != 0:
        raise MemoryError("edmp_nprng_standard_normal: allocation failed")
    np.random.set_state(("MT19937", key, pos.value, has_gauss.value, gauss.value))
    return out.reshape(shape)
