"""ctypes binding of libedmp_hip.so (include/edmp_hip.h).  Fails loudly when the library is missing or does not
load — there is no CPU fallback for the product path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libedmp_hip.so")

MAX_LEVELS = 8


class UNetDesc(C.Structure):
    _fields_ = [
        ("input_dim", C.c_int32),
        ("time_dim", C.c_int32),
        ("n_levels", C.c_int32),
        ("dims", C.c_int32 * MAX_LEVELS),
        ("horizon", C.c_int32),
        ("T", C.c_int32),
    ]


class EdmpError(RuntimeError):
    rc = None  # status code of the failing C-ABI call (include/edmp_hip.h), None for errors raised on the Python side


class EdmpLayoutError(EdmpError):
    """EDMP_ERR_LAYOUT: a packed weight image written by another library version or under other builder switches"""


ERR_LAYOUT = -4


_vp, _i, _d = C.c_void_p, C.c_int, C.c_double
_pd, _pf, _pi32 = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32)

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)  # edmp_allreduce_fn(user, hip_stream, sumsq_dev)

# name -> (restype, argtypes); every symbol include/edmp_hip.h declares
SIGNATURES = {
    "edmp_last_error": (C.c_char_p, []),
    "edmp_version": (_i, []),
    "edmp_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "edmp_ctx_destroy": (None, [_vp]),
    "edmp_ctx_set_stream": (_i, [_vp, _vp]),
    "edmp_ctx_synchronize": (_i, [_vp]),
    "edmp_unet_param_count": (C.c_int64, [C.POINTER(UNetDesc)]),
    "edmp_unet_load": (_i, [_vp, C.POINTER(UNetDesc), _pf, C.c_int64, _i]),
    "edmp_unet_forward_dev": (_i, [_vp, _vp, _i, _i, _vp]),
    "edmp_unet_read_activation_dev": (_i, [_vp, _i, _i, _vp, C.POINTER(_i), C.POINTER(_i)]),
    "edmp_unet_flops": (_i, [_vp, _pd, _pd]),
    "edmp_scene_set": (_i, [_vp, _pd, _i, _pd, _pd, _i, _i, _pf, _pf, _pf]),
    "edmp_rows_set": (_i, [_vp, _pi32, _pf, _pd, _pd, _i, _i]),
    "edmp_scene_read_aabbs": (_i, [_vp, _i, _i, _pf]),
    "edmp_guide_cost_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "edmp_guide_swept_cost_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _pf, _pf, _vp]),
    "edmp_guide_gradient_dev": (_i, [_vp, _vp, _i, _i, _pd, _pd, _i, _vp, _vp]),
    "edmp_row_swept_volumes_dev": (_i, [_vp, _vp, _i, _i, _pd, _pd, _vp, C.POINTER(_i)]),
    "edmp_scene_set_shapes": (_i, [_vp, _pi32, _i]),
    "edmp_success_rows_dev": (_i, [_vp, _vp, _i, _i, _i, _pd, _vp, _vp, _vp, _pi32]),
    "edmp_sampler_init": (_i, [_vp, _i, _d]),
    "edmp_sampler_read_schedule": (_i, [_vp, _pd, _pd, _pd]),
    "edmp_sampler_set_condition": (_i, [_vp, _i]),
    "edmp_psample_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "edmp_step_a_dev": (_i, [_vp, _vp, _vp, _i, _i, _pd, _pd, _i, _vp, _vp]),
    "edmp_step_b_dev": (_i, [_vp, _vp, _i, _i, _pd, _pd, _vp]),
    "edmp_sumsq_ptr_dev": (_vp, [_vp]),
    "edmp_denoise_guided_dev": (_i, [_vp, _vp, _i, _pd, _pd, _i, _i, _i, _vp]),
    "edmp_denoise_guided_segment_dev": (_i, [_vp, _vp, _i, _pd, _pd, _i, _i, _i, _i, _i, _vp]),
    "edmp_denoise_guided_rng_dev": (_i, [_vp, C.c_uint64, _i, _pd, _pd, _i, _i, _i, _vp]),
    "edmp_rng_normal_dev": (_i, [_vp, C.c_uint64, _i, _i, _i, _i, _vp]),
    "edmp_sampler_set_graph": (_i, [_vp, _i]),
    "edmp_q_sample_dev": (_i, [_vp, _vp, _vp, _pi32, _i, _i, _i, _i, _i, _vp, _vp]),
    "edmp_unet_packed_size": (C.c_int64, [_vp, C.POINTER(C.c_int)]),
    "edmp_unet_read_packed": (_i, [_vp, _pf, C.c_int64]),
    "edmp_unet_load_packed": (_i, [_vp, C.POINTER(UNetDesc), _pf, C.c_int64, _i, _i]),
    "edmp_unet_flops_direct": (_i, [_vp, _pd]),
    "edmp_unet_flops_pipes": (_i, [_vp, _pd, _pd]),
    "edmp_prof_ops_bf16": (_i, [_vp, _i, C.POINTER(C.c_int), _pd]),
    "edmp_unet_slot": (_i, [_vp, C.c_uint64]),
    "edmp_guide_slot": (_i, [_vp, C.c_uint64]),
    "edmp_argmin_dev": (_i, [_vp, _vp, _i, C.POINTER(C.c_int)]),
    "edmp_sampler_set_allreduce": (_i, [_vp, _vp, _vp]),
    "edmp_rccl_load": (_i, [C.c_char_p]),
    "edmp_rccl_unique_id": (_i, [_vp]),
    "edmp_rccl_attach": (_i, [_vp, _vp, _i, _i]),
    "edmp_rccl_attach_comm": (_i, [_vp, _vp]),
    "edmp_rccl_enable": (_i, [_vp, _i]),
    "edmp_rccl_detach": (_i, [_vp]),
    "edmp_rccl_info": (_i, [_vp, _pi32]),
    "edmp_sampler_allreduce_stats": (_i, [_vp, C.POINTER(C.c_uint64), _i]),
    "edmp_prof_enable": (_i, [_vp, _i]),
    "edmp_prof_read": (_i, [_vp, _pd, C.POINTER(C.c_int64), _i]),
    "edmp_prof_ops": (_i, [_vp, _i, C.POINTER(C.c_int), _pd, C.POINTER(C.c_int64), _pd, C.c_char_p]),
}

_lib = None


def load():
    """dlopen libedmp_hip.so (built by __graft_entry__.build()).  Raises EdmpError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EdmpError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  edmp_amd has no CPU fallback."
        )
    # PyTorch-ROCm carries its own libamdhip64: map it FIRST, so that the process has ONE HIP runtime (our library then binds to the copy
    # already loaded).  In the other order - this library pulling /opt/rocm's runtime, torch its own afterwards (what
    # `python __graft_entry__.py smoke` did: build() loads the library before anything imported torch) - the process holds two runtimes
    # and the second one reports "no ROCm-capable device".  A host without torch (tests/c_abi) links the system runtime alone.
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover
        pass
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise EdmpError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise EdmpError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().edmp_last_error()
        err = (EdmpLayoutError if rc == ERR_LAYOUT else EdmpError)(f"{what or 'libedmp_hip'} failed (rc={rc}): {msg.decode() if msg else '?'}")
        err.rc = rc
        raise err


def as_pd(a):
    return a.ctypes.data_as(_pd)


def as_pf(a):
    return a.ctypes.data_as(_pf)


def as_pi32(a):
    return a.ctypes.data_as(_pi32)
