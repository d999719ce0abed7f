"""TemporalUNet parameter inventory, seeded initialiser and checkpoint reader.

The reference stores the denoiser as a ``torch.save``-d ``state_dict`` at ``<model_name>/weights_latest.pt`` (+
``losses.npy``) (diffusion/models/temporalunet.py:78-92).  Tensor names/shapes below are those of that state dict
for ``TemporalUNet(input_dim, time_dim, dims)`` (temporalunet.py:11-45, blocks.py) — SURVEY.md §8a "U".

Trained weights are not distributed with the reference, so `init_state_dict` provides a seeded, framework-free
initialiser (NumPy RandomState; uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)) like torch's defaults) used for tests and
benchmarks.  GroupNorm affine parameters are jittered away from (1, 0) so that parity tests exercise them.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np


def unet_param_shapes(input_dim=7, time_dim=32, dims=(32, 64, 128, 256, 512, 512)) -> "OrderedDict[str, tuple]":
    d = [input_dim, *dims]
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["time_embedding.time_mlp.1.weight"] = (time_dim * 4, time_dim)
    s["time_embedding.time_mlp.1.bias"] = (time_dim * 4,)
    s["time_embedding.time_mlp.3.weight"] = (time_dim, time_dim * 4)
    s["time_embedding.time_mlp.3.bias"] = (time_dim,)

    def conv_block(p, cin, cout, k=5):
        s[p + ".block.0.weight"] = (cout, cin, k)
        s[p + ".block.0.bias"] = (cout,)
        s[p + ".block.2.weight"] = (cout,)
        s[p + ".block.2.bias"] = (cout,)

    def rcb(p, cin, cout):
        conv_block(p + ".blocks.0", cin, cout)
        conv_block(p + ".blocks.1", cout, cout)
        s[p + ".time_mlp.time_mlp.1.weight"] = (cout, time_dim)
        s[p + ".time_mlp.time_mlp.1.bias"] = (cout,)
        if cin != cout:
            s[p + ".residual_conv.weight"] = (cout, cin, 1)
            s[p + ".residual_conv.bias"] = (cout,)

    n_down = len(d) - 1
    for i in range(n_down):
        p = f"down_samplers.{i}.down"
        rcb(p + ".0", d[i], d[i + 1])
        rcb(p + ".1", d[i + 1], d[i + 1])
        if i != n_down - 1:
            s[p + ".3.weight"] = (d[i + 1], d[i + 1], 3)
            s[p + ".3.bias"] = (d[i + 1],)
    rcb("middle_block.middle.0", d[-1], d[-1])
    rcb("middle_block.middle.2", d[-1], d[-1])
    for j, i in enumerate(range(len(d) - 1, 1, -1)):  # temporalunet.py:31-32: UpSampler(dims[i-1], dims[i])
        p = f"up_samplers.{j}.up"
        rcb(p + ".0", d[i] * 2, d[i - 1])
        rcb(p + ".1", d[i - 1], d[i - 1])
        s[p + ".3.weight"] = (d[i - 1], d[i - 1], 4)  # ConvTranspose1d weight is (Cin, Cout, k)
        s[p + ".3.bias"] = (d[i - 1],)
    conv_block("final_conv.0", d[1], d[1])
    s["final_conv.1.weight"] = (input_dim, d[1], 1)
    s["final_conv.1.bias"] = (input_dim,)
    return s


def init_state_dict(seed=0, input_dim=7, time_dim=32, dims=(32, 64, 128, 256, 512, 512), gn_jitter=True):
    """name -> float32 ndarray."""
    rs = np.random.RandomState(seed)
    shapes = unet_param_shapes(input_dim, time_dim, dims)
    sd = OrderedDict()
    for name, shp in shapes.items():
        is_gn = ".block.2." in name
        if is_gn:
            if name.endswith("weight"):
                v = rs.uniform(0.5, 1.5, shp) if gn_jitter else np.ones(shp)
            else:
                v = rs.uniform(-0.2, 0.2, shp) if gn_jitter else np.zeros(shp)
        else:
            if name.endswith("weight"):
                if ".3.weight" in name and name.startswith("up_samplers") and len(shp) == 3:
                    fan_in = shp[1] * shp[2]  # torch: fan_in of a ConvTranspose weight is size(1)*k
                else:
                    fan_in = int(np.prod(shp[1:]))
            else:
                wshape = shapes[name[: -len("bias")] + "weight"]
                if ".3." in name and name.startswith("up_samplers") and len(wshape) == 3:
                    fan_in = wshape[1] * wshape[2]
                else:
                    fan_in = int(np.prod(wshape[1:]))
            bound = 1.0 / np.sqrt(fan_in)
            v = rs.uniform(-bound, bound, shp)
        sd[name] = np.ascontiguousarray(v, dtype=np.float32)
    return sd


def infer_dims(sd) -> tuple:
    """(input_dim, time_dim, dims) from a state dict's shapes."""
    time_dim = int(np.shape(sd["time_embedding.time_mlp.1.weight"])[1])
    n_down = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("down_samplers."))
    input_dim = int(np.shape(sd["down_samplers.0.down.0.blocks.0.block.0.weight"])[1])
    dims = tuple(int(np.shape(sd[f"down_samplers.{i}.down.0.blocks.0.block.0.weight"])[0]) for i in range(n_down))
    return input_dim, time_dim, dims


def load_checkpoint_dir(model_name: str):
    """Read ``<model_name>/weights_latest.pt`` (temporalunet.py:88-92) -> name -> float32 ndarray."""
    import torch

    path = os.path.join(model_name, "weights_latest.pt")
    sd = torch.load(path, map_location="cpu")
    return OrderedDict((k, v.detach().to(torch.float32).cpu().numpy()) for k, v in sd.items())


def save_checkpoint_dir(model_name: str, sd) -> None:
    """Write a state dict in the reference's on-disk format (so either implementation can load it)."""
    import torch

    os.makedirs(model_name, exist_ok=True)
    torch.save(OrderedDict((k, torch.from_numpy(np.asarray(v))) for k, v in sd.items()), os.path.join(model_name, "weights_latest.pt"))
    np.save(os.path.join(model_name, "losses.npy"), np.array([]))


# ---- packed weight image (edmp_unet_load_packed) --------------------------------------------------------------------
PACKED_NAME = "weights_packed.edmp"
_PACK_MAGIC = b"EDMPWPK1"
_PACK_HEADER = 4096  # the float blob starts page-aligned so that it can be mmap'ed and handed to hipMemcpy as is


def write_packed(path: str, layout: int, input_dim: int, time_dim: int, dims, horizon: int, T: int, blob: np.ndarray) -> None:
    """``blob``: float32 device image returned by edmp_unet_read_packed."""
    import struct

    dims = tuple(int(d) for d in dims)
    head = _PACK_MAGIC + struct.pack("<5i8i2iq", int(layout), int(input_dim), int(time_dim), len(dims), 0, *(dims + (0,) * (8 - len(dims))), int(horizon), int(T), int(blob.size))
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(head.ljust(_PACK_HEADER, b"\0"))
        f.write(np.ascontiguousarray(blob, dtype="<f4").tobytes())
    os.replace(tmp, path)


def read_packed(path: str):
    """-> dict(layout, input_dim, time_dim, dims, horizon, T, blob (read-only float32 memmap)) or None if not a packed file."""
    import struct

    try:
        with open(path, "rb") as f:
            head = f.read(_PACK_HEADER)
    except OSError:
        return None
    if len(head) < _PACK_HEADER or head[:8] != _PACK_MAGIC:
        return None
    vals = struct.unpack("<5i8i2iq", head[8 : 8 + struct.calcsize("<5i8i2iq")])
    layout, input_dim, time_dim, n_levels, _ = vals[:5]
    dims = tuple(vals[5 : 5 + n_levels])
    horizon, T, n = vals[13], vals[14], vals[15]
    if os.path.getsize(path) != _PACK_HEADER + 4 * n:
        return None
    blob = np.memmap(path, dtype="<f4", mode="r", offset=_PACK_HEADER, shape=(n,))
    return dict(layout=layout, input_dim=input_dim, time_dim=time_dim, dims=dims, horizon=horizon, T=T, blob=blob)
