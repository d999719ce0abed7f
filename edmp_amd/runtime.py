"""Per-GPU context: owns the libedmp_hip handle, a torch stream for copies + kernels, and tracks which model / guide
object is currently bound.  PyTorch is plumbing here (device memory, streams, torch.distributed) — all arithmetic of
the hot path runs in libedmp_hip.so."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi

_contexts: dict = {}


def _device_index(device) -> int:
    if isinstance(device, int):
        return device
    d = torch.device(device)
    if d.type != "cuda":
        raise _capi.EdmpError(
            f"edmp_amd runs on an MI355X only (device={device!r}).  There is no CPU path: use device='cuda:<i>'."
        )
    return d.index if d.index is not None else torch.cuda.current_device()


class Context:
    def __init__(self, index: int):
        if not torch.cuda.is_available():
            raise _capi.EdmpError("no GPU visible to torch: edmp_amd needs an MI355X (gfx950); there is no CPU fallback")
        self.lib = _capi.load()
        self.index = index
        self.device = torch.device("cuda", index)
        h = C.c_void_p()
        _capi.check(self.lib.edmp_ctx_create(index, C.byref(h)), "edmp_ctx_create")
        self.h = h
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.Stream(device=self.device)
        _capi.check(self.lib.edmp_ctx_set_stream(self.h, C.c_void_p(self.stream.cuda_stream)), "edmp_ctx_set_stream")
        self.bound_model = None
        self.bound_guide = None
        self.sampler_T = None

    def close(self):
        """release everything the context holds on the GPU and the host: edmp_ctx_destroy (resident models and guides, activation
        buffers, sampler state, streams), the pinned staging ring and the draw thread.  Objects bound to a closed context must not
        be used again.  Contexts obtained from get_context / lane_context are cached for the life of the process and normally
        never closed; a caller that creates `Context(i)` itself owns it and closes it."""
        if getattr(self, "h", None) is None:
            return
        pool = getattr(self, "_draw_pool", None)
        if pool is not None:
            pool.shutdown(wait=True)
            self._draw_pool = None
        try:
            self.stream.synchronize()
        finally:
            self.lib.edmp_ctx_destroy(self.h)
            self.h = None
            self.bound_model = self.bound_guide = None
            self._pinned = None
            self._scene_noise_buffers = None  # (infer_serial's whole-scene pinned noise buffers of scenes in flight)
            for key in [k for k, v in _contexts.items() if v is self] + [k for k, v in _lane_contexts.items() if v is self]:
                _contexts.pop(key, None)
                _lane_contexts.pop(key, None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # ---- helpers ---------------------------------------------------------------------------------------------
    def to_dev(self, arr, dtype) -> torch.Tensor:
        """host ndarray / tensor -> contiguous device tensor on this context's stream."""
        if isinstance(arr, torch.Tensor) and arr.is_cuda:
            self.adopt(arr)  # produced on the caller's stream
        with torch.cuda.stream(self.stream):
            if isinstance(arr, torch.Tensor):
                t = arr.to(device=self.device, dtype=dtype).contiguous()
            else:
                t = torch.from_numpy(np.ascontiguousarray(arr)).to(dtype=dtype).to(self.device)
        return t

    def to_dev_overlapped(self, arr: np.ndarray, dtype) -> torch.Tensor:
        """host ndarray -> device tensor through a SEPARATE copy stream (the DMA engine then runs beside the kernels of
        this context's stream instead of in line with them); this context's stream is ordered after the copy.  The caller
        keeps the tensor alive until the work that reads it has been enqueued AND completed (or synchronises)."""
        if not hasattr(self, "copy_stream"):
            with torch.cuda.device(self.device):
                self.copy_stream = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(self.copy_stream):
            t = torch.from_numpy(np.ascontiguousarray(arr)).to(dtype=dtype).to(self.device)
        t.record_stream(self.stream)
        self.stream.wait_stream(self.copy_stream)
        return t

    def pinned_ring(self, n_slots: int, n_doubles: int):
        """ring of page-locked float64 staging buffers (torch pinned tensors + their NumPy views), grown on demand and kept for
        the life of the context: hipHostMalloc of ~50 MB costs milliseconds, far more than a chunk's upload."""
        ring = getattr(self, "_pinned", None)
        if ring is None or len(ring) < n_slots or ring[0]["t"].numel() < n_doubles:
            ring = []
            for _ in range(n_slots):
                t = torch.empty(int(n_doubles), dtype=torch.float64, pin_memory=True)
                ring.append({"t": t, "np": t.numpy(), "event": None})
            self._pinned = ring
        return ring

    def draw_pool(self):
        """the context's single background thread for host-side noise draws.  Long-lived on purpose: libgomp keeps one thread
        team per master thread, so a fresh executor per scene would create (and place) a new team for its first draw - measured
        18-80 ms on the GPU box, against 1.3 ms for the draw itself."""
        if getattr(self, "_draw_pool", None) is None:
            from concurrent.futures import ThreadPoolExecutor

            self._draw_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="edmp-draw")
        return self._draw_pool

    def upload_pinned(self, slot, n: int) -> torch.Tensor:
        """first n doubles of a pinned staging buffer -> a fresh device tensor by asynchronous DMA on the copy stream; this
        context's stream is ordered after the copy, the slot's event marks when the buffer may be overwritten."""
        if not hasattr(self, "copy_stream"):
            with torch.cuda.device(self.device):
                self.copy_stream = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(self.copy_stream):
            d = torch.empty(int(n), dtype=torch.float64, device=self.device)
            d.copy_(slot["t"][:n], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        slot["event"] = ev
        d.record_stream(self.stream)
        self.stream.wait_stream(self.copy_stream)
        return d

    def empty(self, shape, dtype) -> torch.Tensor:
        with torch.cuda.stream(self.stream):
            return torch.empty(shape, dtype=dtype, device=self.device)

    def adopt(self, t: torch.Tensor) -> torch.Tensor:
        """a device tensor produced by the CALLER (on torch's current stream): order this context's stream after it."""
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        return t

    def hand_over(self, t: torch.Tensor) -> torch.Tensor:
        """a device tensor produced on this context's stream, about to be returned: order torch's current stream after
        it, so the caller's torch ops on it are safe without an explicit synchronisation."""
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        return t

    def sync(self):
        _capi.check(self.lib.edmp_ctx_synchronize(self.h), "edmp_ctx_synchronize")

    def to_host(self, t: torch.Tensor) -> np.ndarray:
        with torch.cuda.stream(self.stream):
            out = t.cpu()
        self.stream.synchronize()
        return out.numpy()

    def ensure_sampler(self, T: int, variance_thresh: float = 0.02):
        key = (T, variance_thresh)
        if self.sampler_T != key:
            _capi.check(self.lib.edmp_sampler_init(self.h, int(T), float(variance_thresh)), "edmp_sampler_init")
            self.sampler_T = key

    def prof(self, on):
        """0/False off, 1/True per-launch event brackets, 2 one bracket around the whole UNet layer program."""
        _capi.check(self.lib.edmp_prof_enable(self.h, int(on)))

    def prof_read(self, reset=True):
        ms = C.c_double()
        n = C.c_int64()
        _capi.check(self.lib.edmp_prof_read(self.h, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    def prof_ops(self):
        """per program op of the loaded UNet: (kernel instance name, launches, summed event ms, executed FLOPs per
        trajectory per launch) since the last prof_read(reset=True); call BEFORE that reset."""
        cap = 256
        n = C.c_int()
        ms = (C.c_double * cap)()
        calls = (C.c_int64 * cap)()
        fl = (C.c_double * cap)()
        names = C.create_string_buffer(cap * 64)
        _capi.check(self.lib.edmp_prof_ops(self.h, cap, C.byref(n), ms, calls, fl, names))
        out = []
        for i in range(min(n.value, cap)):
            nm = names.raw[i * 64:(i + 1) * 64].split(b"\0", 1)[0].decode()
            out.append((nm, int(calls[i]), float(ms[i]), float(fl[i])))
        return out

    def prof_ops_bf16(self):
        """per program op: FLOPs per trajectory per launch issued on the bf16 matrix pipe (bf16x3 ops; 0 for fp32-MFMA ops)."""
        cap = 256
        n = C.c_int()
        fl = (C.c_double * cap)()
        _capi.check(self.lib.edmp_prof_ops_bf16(self.h, cap, C.byref(n), fl))
        return [float(fl[i]) for i in range(min(n.value, cap))]


_slot_counter = [0]


def new_slot_key() -> int:
    """process-unique non-zero key of a model / guide object's resident slot (edmp_unet_slot / edmp_guide_slot)."""
    _slot_counter[0] += 1
    return _slot_counter[0]


def get_context(device) -> Context:
    """the per-GPU context of `device` ("cuda:<i>").  A `Context` instance passes through: objects built on an explicitly
    created second context (`Context(i)`) live on their own stream with their own resident model / guides - two scenes in
    flight on one GPU (scripts/two_scenes.py)."""
    if isinstance(device, Context):
        return device
    idx = _device_index(device)
    if idx not in _contexts:
        _contexts[idx] = Context(idx)
    return _contexts[idx]


_lane_contexts: dict = {}


def lane_context(device, lane: int) -> Context:
    """the context of lane `lane` of a GPU: lane 0 is get_context(device), lanes >= 1 are further contexts of the same device
    (own stream, own resident model and guides) for several scenes in flight.  Cached per (device, lane): repeated
    infer_serial.run(scenes_in_flight=k) calls reuse the lanes instead of leaking a resident UNet, activation buffers, streams
    and a pinned ring per call."""
    base = get_context(device)
    if lane <= 0:
        return base
    key = (base.index, int(lane))
    if key not in _lane_contexts:
        _lane_contexts[key] = Context(base.index)
    return _lane_contexts[key]


def ptr(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())
