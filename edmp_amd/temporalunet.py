"""TemporalUNet — host-side mirror of the reference denoiser object (diffusion/models/temporalunet.py:9-100) whose
forward runs in libedmp_hip.so (edmp_amd/csrc/unet.hip)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _capi, weights
from .runtime import get_context, new_slot_key, ptr


class TemporalUNet:
    """Same constructor / call signature as the reference:

        net = TemporalUNet(model_name, input_dim, time_dim, device, dims=(32, 64, 128, 256))
        eps = net(x, t)          # x (B, input_dim, 50) f32, t (1,) f32  ->  (B, input_dim, 50) f32 tensor on `device`
        net.train(False)

    Reference semantics kept: if the directory ``model_name`` does not exist it is created and the network keeps
    a random initialisation (temporalunet.py:39-41), otherwise ``weights_latest.pt`` is loaded (:42-43, :88-92).
    Extras (keyword only): ``state_dict`` (name -> array, overrides disk), ``seed`` for the random init,
    ``max_batch`` (activation workspace size), ``horizon`` (50), ``T`` (size of the precomputed time-bias table).
    """

    def __init__(self, model_name, input_dim, time_dim, device, dims=(32, 64, 128, 256), *, state_dict=None, seed=0,
                 max_batch=1024, horizon=50, T=255, use_packed=True):
        self.model_name = model_name
        self.input_dim, self.time_dim, self.dims = int(input_dim), int(time_dim), tuple(int(d) for d in dims)
        self.horizon, self.T = int(horizon), int(T)
        self.max_batch = int(max_batch)
        self.ctx = get_context(device)
        self.device = self.ctx.device
        self._flat = None
        self._packed = None
        self._slot = new_slot_key()
        if state_dict is None and model_name is not None and os.path.exists(model_name) and use_packed:
            # fast path: the device image written by pack() - one mmap + one host-to-device copy instead of torch.load,
            # 290 tensor copies and the host-side repack (used only if it matches this architecture, this library's
            # packing layout, and is not older than the state-dict checkpoint)
            pk = weights.read_packed(os.path.join(model_name, weights.PACKED_NAME))
            ck = os.path.join(model_name, "weights_latest.pt")
            if (pk is not None and (pk["input_dim"], pk["time_dim"], pk["dims"], pk["horizon"], pk["T"]) ==
                    (self.input_dim, self.time_dim, self.dims, self.horizon, self.T)
                    and (not os.path.exists(ck) or os.path.getmtime(os.path.join(model_name, weights.PACKED_NAME)) >= os.path.getmtime(ck))):
                # the image's layout id covers the library's packing version AND the layout the builder produces under the
                # current run-time switches (EDMP_NO_*): edmp_unet_load_packed recomputes it and refuses a mismatch - then
                # the state-dict checkpoint is loaded instead
                self._packed = pk
                try:
                    self._bind()
                except _capi.EdmpLayoutError as e:
                    # ONLY a layout mismatch falls back to the state dict (stale image / other builder switches); an allocation
                    # or HIP failure is a real error and propagates instead of being retried and masked
                    self._packed = None
                    if not os.path.exists(ck):
                        raise
                    print(f"[edmp_amd] {os.path.join(model_name, weights.PACKED_NAME)} does not match this library's packing ({e}); loading "
                          f"weights_latest.pt instead - re-pack with TemporalUNet.pack()")
                else:
                    self.losses = np.load(os.path.join(model_name, "losses.npy")) if os.path.exists(os.path.join(model_name, "losses.npy")) else np.array([])
                    print("Loaded Model at " + str(self.losses.size) + " epochs")
                    return
        if state_dict is None:
            if model_name is not None and os.path.exists(model_name):
                state_dict = weights.load_checkpoint_dir(model_name)
                self.losses = np.load(os.path.join(model_name, "losses.npy")) if os.path.exists(os.path.join(model_name, "losses.npy")) else np.array([])
                print("Loaded Model at " + str(self.losses.size) + " epochs")
            else:
                if model_name is not None:
                    os.mkdir(model_name)
                self.losses = np.array([])
                state_dict = weights.init_state_dict(seed, self.input_dim, self.time_dim, self.dims)
        self._flat = self._flatten(state_dict)
        self._bind()

    def _flatten(self, state_dict):
        shapes = weights.unet_param_shapes(self.input_dim, self.time_dim, self.dims)
        missing = [k for k in shapes if k not in state_dict]
        if missing:
            raise KeyError(f"state dict lacks {len(missing)} tensors, e.g. {missing[:3]}")
        flat = []
        for k, shp in shapes.items():
            v = np.asarray(state_dict[k].detach().cpu().numpy() if isinstance(state_dict[k], torch.Tensor) else state_dict[k], dtype=np.float32)
            if tuple(v.shape) != tuple(shp):
                raise ValueError(f"{k}: shape {tuple(v.shape)} != expected {tuple(shp)}")
            flat.append(v.reshape(-1))
        return np.ascontiguousarray(np.concatenate(flat))

    def _desc(self):
        d = _capi.UNetDesc()
        d.input_dim, d.time_dim, d.n_levels = self.input_dim, self.time_dim, len(self.dims)
        for i, v in enumerate(self.dims):
            d.dims[i] = v
        d.horizon, d.T = self.horizon, self.T
        return d

    def _bind(self):
        ctx = self.ctx
        if ctx.bound_model is self:
            return
        # switch the context to this object's resident slot; only an empty slot (first use, or evicted) uploads the weights
        have = ctx.lib.edmp_unet_slot(ctx.h, self._slot)
        if have < 0:
            _capi.check(have, "edmp_unet_slot")
        if have == 1:
            ctx.bound_model = self
            return
        ctx.bound_model = None
        d = self._desc()
        if self._packed is not None:
            blob = self._packed["blob"]
            _capi.check(ctx.lib.edmp_unet_load_packed(ctx.h, C.byref(d), C.cast(blob.ctypes.data, _capi._pf), blob.size, self._packed["layout"], self.max_batch),
                        "edmp_unet_load_packed")
        else:
            n = ctx.lib.edmp_unet_param_count(C.byref(d))
            if n != self._flat.size:
                raise _capi.EdmpError(f"parameter count mismatch: python {self._flat.size}, library {n}")
            _capi.check(ctx.lib.edmp_unet_load(ctx.h, C.byref(d), _capi.as_pf(self._flat), self._flat.size, self.max_batch), "edmp_unet_load")
        ctx.bound_model = self

    def train(self, mode=True):
        return self

    def eval(self):
        return self

    def forward(self, x, t):
        ctx = self.ctx
        self._bind()
        tt = float(t.reshape(-1)[0]) if isinstance(t, (torch.Tensor, np.ndarray)) else float(t)
        ti = int(round(tt))
        if ti != tt:
            raise ValueError(f"t must be an integer diffusion step (got {tt}); the time-bias table is precomputed for t=1..{self.T}")
        xd = ctx.to_dev(x, torch.float32)
        if xd.dim() != 3 or xd.shape[1] != self.input_dim or xd.shape[2] != self.horizon:
            raise ValueError(f"x must be (B, {self.input_dim}, {self.horizon}), got {tuple(xd.shape)}")
        eps = ctx.empty(xd.shape, torch.float32)
        _capi.check(ctx.lib.edmp_unet_forward_dev(ctx.h, ptr(xd), xd.shape[0], ti, ptr(eps)), "edmp_unet_forward_dev")
        ctx.sync()
        return eps

    __call__ = forward

    def activation(self, which: int, B: int):
        """(B, C, L) f32 copy of an internal activation of THIS model's last forward (parity/debug)."""
        ctx = self.ctx
        self._bind()  # (every resident model keeps its own activation buffers)
        buf = ctx.empty((B * 4096,), torch.float32)
        c, l = C.c_int(), C.c_int()
        _capi.check(ctx.lib.edmp_unet_read_activation_dev(ctx.h, which, B, ptr(buf), C.byref(c), C.byref(l)))
        ctx.sync()
        return buf[: B * c.value * l.value].reshape(B, c.value, l.value)

    def flops_per_trajectory(self):
        self._bind()
        a, b = C.c_double(), C.c_double()
        _capi.check(self.ctx.lib.edmp_unet_flops(self.ctx.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def flops_by_pipe(self):
        """(fp32-pipe, bf16-pipe) MFMA FLOPs issued per trajectory per forward (see edmp_unet_flops_pipes)."""
        self._bind()
        a, b = C.c_double(), C.c_double()
        _capi.check(self.ctx.lib.edmp_unet_flops_pipes(self.ctx.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def pack(self, path=None):
        """Write the device weight image next to the checkpoint (<model_name>/weights_packed.edmp, or ``path``): later
        constructions of this architecture load it with one mmap + one copy.  No reference counterpart."""
        self._bind()
        ctx = self.ctx
        layout = C.c_int()
        n = ctx.lib.edmp_unet_packed_size(ctx.h, C.byref(layout))
        blob = np.empty(n, dtype=np.float32)
        _capi.check(ctx.lib.edmp_unet_read_packed(ctx.h, _capi.as_pf(blob), n), "edmp_unet_read_packed")
        if path is None:
            if self.model_name is None:
                raise ValueError("pack() needs a path for a model without a directory")
            path = os.path.join(self.model_name, weights.PACKED_NAME)
        weights.write_packed(path, layout.value, self.input_dim, self.time_dim, self.dims, self.horizon, self.T, blob)
        return path

    def flops_direct_form(self):
        """FLOPs per trajectory of the direct convolution without padding taps (see edmp_unet_flops_direct)."""
        self._bind()
        d = C.c_double()
        _capi.check(self.ctx.lib.edmp_unet_flops_direct(self.ctx.h, C.byref(d)))
        return d.value

    def save(self):
        if self._flat is None:  # constructed from a packed image: the state dict lives in the checkpoint next to it
            self._flat = self._flatten(weights.load_checkpoint_dir(self.model_name))
        sd = {}
        off = 0
        for k, shp in weights.unet_param_shapes(self.input_dim, self.time_dim, self.dims).items():
            n = int(np.prod(shp))
            sd[k] = self._flat[off : off + n].reshape(shp)
            off += n
        weights.save_checkpoint_dir(self.model_name, sd)
