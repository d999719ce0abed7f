"""Import the UNMODIFIED reference (/root/reference) in the build container — TEST INFRASTRUCTURE ONLY.

Used by ``oracle/gen_golden.py`` to pin ``oracle/edmp_oracle.py`` and to emit ``tests/golden/*.npz``.  Never runs on
the GPU box (the reference does not travel).  Absent third-party modules are replaced by minimal stand-ins that
only make ``import lib`` / ``import diffusion`` succeed; none of them implements arithmetic on the hot path:

* pybullet, pybullet_utils.bullet_client, h5py, wandb, robofin.robots — imported by lib/environment.py:1-13 only.
* autolab_core.YamlConfig -> dict(yaml.safe_load)            (infer_serial.py:25,73)
* torchvision.transforms.functional.crop -> tensor slicing     (temporalunet.py:71)
* pybullet_data.getDataPath() -> temp dir with 8-vertex box .obj files whose extents are the chosen link table
  (lib/guide.py:245-269 only takes max-min of the 'v' lines).
* NumPy-2 shim for quirk Q3: single-argument np.where on a 0-d condition returns NumPy-1.x's result.
"""
from __future__ import annotations

import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
LINK_NAMES = ["link1", "link2", "link3", "link4", "link5", "link6", "link7", "hand", "finger"]


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF, "lib"))


def _write_box_obj(path, ext):
    hx, hy, hz = (e / 2 for e in ext)
    with open(path, "w") as f:
        for sx in (-1, 1):
            for sy in (-1, 1):
                for sz in (-1, 1):
                    f.write(f"v {sx * hx:.17g} {sy * hy:.17g} {sz * hz:.17g}\n")


def install(link_mesh_extents):
    """Install stand-in modules + mesh dir, put the reference on sys.path.  Returns the imported modules."""
    datadir = tempfile.mkdtemp(prefix="edmp_pbdata_")
    mesh = os.path.join(datadir, "franka_panda", "meshes", "collision")
    os.makedirs(mesh)
    for name, ext in zip(LINK_NAMES, link_mesh_extents):
        _write_box_obj(os.path.join(mesh, name + ".obj"), ext)

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("pybullet", GUI=1, DIRECT=2)
    pu = mod("pybullet_utils")
    pu.bullet_client = mod("pybullet_utils.bullet_client", BulletClient=object)
    mod("pybullet_data", getDataPath=lambda: datadir)
    mod("h5py")
    mod("wandb")
    rf = mod("robofin")
    rf.robots = mod("robofin.robots", FrankaRobot=object)
    import yaml

    def YamlConfig(path):
        with open(path) as f:
            return dict(yaml.safe_load(f))

    mod("autolab_core", YamlConfig=YamlConfig)
    tv = mod("torchvision")
    tvt = mod("torchvision.transforms")
    tvf = mod("torchvision.transforms.functional", crop=lambda x, top, left, h, w: x[..., top : top + h, left : left + w])
    tv.transforms = tvt
    tvt.functional = tvf

    _orig_where = np.where

    def where_np1(cond, *args):  # NumPy-1.x semantics for np.where(<0-d>)
        if not args and np.ndim(cond) == 0:
            return np.atleast_1d(cond).nonzero()
        return _orig_where(cond, *args)

    np.where = where_np1

    if REF not in sys.path:
        sys.path.insert(0, REF)
    import diffusion as ref_diffusion  # noqa: E402
    import lib.guide as ref_guide  # noqa: E402

    return ref_diffusion, ref_guide
