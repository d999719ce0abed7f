"""Franka Panda kinematic / collision-box tables handed to the HIP guide kernels.

Values follow the reference's IntersectionVolumeGuide: modified-DH rows (lib/guide.py:29-38), link -> joint-frame
map (lib/guide.py:93-94, 286), static link-box frames (lib/guide.py:289-340), corner order (lib/guide.py:210-235)
and the joint limits used to clip the guide input (diffusion/diffusion.py:282-296).

Link-box extents: the reference measures them from pybullet_data's Franka collision meshes every time a guide is
built (lib/guide.py:245-282).  `link_extents_from_mesh_dir` is that reader; `resolve_link_extents` is the lookup order
of every guide built here: an explicit (9,3) table, else a mesh directory (`mesh_dir=` / the run config's
`model.mesh_dir`), else pybullet_data's directory when that package is importable, else `PLACEHOLDER_LINK_EXTENTS` - a
documented stand-in for offline boxes - with ONE warning per process, because results then differ from the reference's.
"""
from __future__ import annotations

import math
import os
import warnings

import numpy as np

N_JOINTS = 7
N_LINKS = 9
LINK_NAMES = ("link1", "link2", "link3", "link4", "link5", "link6", "link7", "hand", "finger")

_H = math.pi / 2
# [a, d, alpha] per joint; theta = q_i
DH_A_D_ALPHA = np.array(
    [[0, 0.333, 0], [0, 0, -_H], [0, 0.316, _H], [0.0825, 0, _H], [-0.0825, 0.384, -_H], [0, 0, _H], [0.088, 0, _H]],
    dtype=np.float64,
)
LINK_FRAME = np.array([0, 1, 2, 3, 4, 5, 6, 6, 6], dtype=np.int32)

_C, _S = 7.07106767e-01, 7.07106795e-01
_TRANS = [
    (8.71e-05, -3.709035e-02, -6.851545e-02),
    (-8.425e-05, -6.93950016e-02, 3.71961970e-02),
    (0.0414576, 0.0281429, -0.03293086),
    (-4.12337575e-02, 3.44296512e-02, 2.79226985e-02),
    (3.345e-05, 3.738805e-02, -1.0619285e-01),
    (4.21935e-02, 1.52195003e-02, 6.07699933e-03),
    (1.863575e-02, 1.85788569e-02, 7.94137484e-02),
    (-1.26717073e-03, -1.25294673e-03, 1.27018693e-01),
    (9.29352476e-03, 9.28272434e-03, 1.92390375e-01),
]


def static_frames() -> np.ndarray:
    """(9, 3, 4) float32 [R | t] of each link box in its joint frame (links 7, 8 are yawed -45 deg)."""
    f = np.zeros((N_LINKS, 3, 4), dtype=np.float64)
    for i, t in enumerate(_TRANS):
        f[i, :, :3] = np.eye(3)
        if i >= 7:
            f[i, :2, :2] = [[_C, _S], [-_S, _C]]
        f[i, :, 3] = t
    return f.astype(np.float32)


def dh_table() -> np.ndarray:
    """(7, 4) float32 [a, d, cos(alpha), sin(alpha)], the trig evaluated in float32 like the reference does
    (torch.cos / torch.sin on the float32 alpha, lib/guide.py:59-67), so cos(pi/2) is -4.37e-8, not 0."""
    al = DH_A_D_ALPHA[:, 2].astype(np.float32)
    t = np.zeros((N_JOINTS, 4), dtype=np.float32)
    t[:, 0] = DH_A_D_ALPHA[:, 0]
    t[:, 1] = DH_A_D_ALPHA[:, 1]
    t[:, 2] = np.cos(al)
    t[:, 3] = np.sin(al)
    return t


def dh_table_f64() -> np.ndarray:
    """(7, 4) float64 [a, d, cos(alpha), sin(alpha)]: the success check walks the chain in float64."""
    t = np.zeros((N_JOINTS, 4), dtype=np.float64)
    t[:, 0] = DH_A_D_ALPHA[:, 0]
    t[:, 1] = DH_A_D_ALPHA[:, 1]
    t[:, 2] = np.cos(DH_A_D_ALPHA[:, 2])
    t[:, 3] = np.sin(DH_A_D_ALPHA[:, 2])
    return t


JOINT_LOWER_DEG = (-166.0, -101.0, -166.0, -176.0, -166.0, -1.0, -166.0)
JOINT_UPPER_DEG = (166.0, 101.0, 166.0, -4.0, 166.0, 215.0, 166.0)


def joint_limits():
    """float64 (lower, upper) in rad, computed as deg*(pi/180) like diffusion.py:282-296."""
    lo = np.array([d * (np.pi / 180) for d in JOINT_LOWER_DEG])
    hi = np.array([d * (np.pi / 180) for d in JOINT_UPPER_DEG])
    return lo, hi


# PLACEHOLDER (see module docstring): (l, b, h) mesh AABB extents for link1..link7, hand, finger, finger y BEFORE x4
PLACEHOLDER_LINK_EXTENTS = np.array(
    [
        [0.110, 0.174, 0.260],
        [0.110, 0.260, 0.175],
        [0.180, 0.170, 0.190],
        [0.180, 0.175, 0.170],
        [0.110, 0.190, 0.360],
        [0.185, 0.140, 0.115],
        [0.110, 0.110, 0.095],
        [0.065, 0.205, 0.095],
        [0.022, 0.016, 0.055],
    ],
    dtype=np.float64,
)


MESH_SUBDIR = os.path.join("franka_panda", "meshes", "collision")  # below pybullet_data.getDataPath() (lib/guide.py:245)


def link_extents_from_mesh_dir(path) -> np.ndarray:
    """(9, 3) float64 AABB extents (max - min over every vertex) of `<path>/{link1..7,hand,finger}.obj`, read the way
    the reference does (lib/guide.py:245-269): a vertex is a line that, stripped, starts with 'v ' (so 'vn' / 'vt' / 'f' /
    comments are skipped), its coordinates are the first three whitespace-separated fields after the 'v'.  The finger's
    y x 4 (lib/guide.py:278-279) is NOT applied here: `link_half_extents` does that, for tables from any source."""
    path = os.fspath(path)
    ext = np.zeros((N_LINKS, 3), dtype=np.float64)
    for i, name in enumerate(LINK_NAMES):
        fn = os.path.join(path, name + ".obj")
        if not os.path.isfile(fn):
            raise FileNotFoundError(f"link mesh {fn} not found (expected {', '.join(n + '.obj' for n in LINK_NAMES)} in {path})")
        verts = []
        with open(fn, "r") as f:
            for line in f:
                line = line.strip()
                if line.startswith("v "):
                    verts.append([float(c) for c in line.split()[1:4]])
        if not verts:
            raise ValueError(f"{fn}: no vertex ('v x y z') lines")
        v = np.array(verts, dtype=np.float64)
        ext[i] = np.max(v, axis=0) - np.min(v, axis=0)
    return ext


_warned_placeholder = False


def default_mesh_dir():
    """pybullet_data's Franka collision-mesh directory if that package is importable (the reference's source,
    lib/guide.py:245), else None."""
    try:
        import pybullet_data  # noqa: PLC0415 (optional third-party data package)
    except Exception:
        return None
    d = os.path.join(pybullet_data.getDataPath(), MESH_SUBDIR)
    return d if os.path.isdir(d) else None


def resolve_link_extents(link_mesh_extents=None, mesh_dir=None) -> np.ndarray:
    """The (9, 3) mesh-extent table a guide is built with: explicit table > `mesh_dir` > pybullet_data > placeholder (warns once)."""
    global _warned_placeholder
    if link_mesh_extents is not None:
        ext = np.array(link_mesh_extents, dtype=np.float64)
        if ext.shape != (N_LINKS, 3):
            raise ValueError(f"link_mesh_extents must be (9, 3), got {ext.shape}")
        return ext
    if mesh_dir is None:
        mesh_dir = os.environ.get("EDMP_MESH_DIR") or default_mesh_dir()
    if mesh_dir is not None:
        return link_extents_from_mesh_dir(mesh_dir)
    if not _warned_placeholder:
        _warned_placeholder = True
        warnings.warn("edmp_amd: no Franka collision meshes found (pybullet_data not importable, no mesh_dir / model.mesh_dir / EDMP_MESH_DIR): "
                      "link boxes use franka.PLACEHOLDER_LINK_EXTENTS - results differ from the reference's, which measures "
                      "pybullet_data/franka_panda/meshes/collision/*.obj (lib/guide.py:245-282)", RuntimeWarning, stacklevel=3)
    return PLACEHOLDER_LINK_EXTENTS.copy()


def link_half_extents(link_mesh_extents=None) -> np.ndarray:
    """(9, 3) float32 half extents; applies the reference's finger y x4 (lib/guide.py:278-279) in float64, casts
    to float32 (lib/guide.py:282) and halves in float32 (lib/guide.py:210-235)."""
    ext = np.array(PLACEHOLDER_LINK_EXTENTS if link_mesh_extents is None else link_mesh_extents, dtype=np.float64)
    if ext.shape != (N_LINKS, 3):
        raise ValueError(f"link_mesh_extents must be (9, 3), got {ext.shape}")
    ext = ext.copy()
    ext[-1, 1] *= 4
    return ext.astype(np.float32) / np.float32(2)
