"""Result evaluation on the host (NumPy): geometric success check and trajectory metrics.

The reference scores a plan by executing it in pybullet (`RobotEnvironment.benchmark_trajectory`,
lib/environment.py:632-680: position control through the waypoints, contact query `check_collisions` :591-608) and
has path-length / SPARC helpers in lib/metrics.py:11-125 (never called by the driver).  pybullet is not available
offline, so the success criterion is restated geometrically — EXACT oriented-box tests instead of the guide's
conservative world-AABB overlap: the 9 Franka link boxes (lib/guide.py:243-342) against every obstacle box at every
waypoint and at `substeps` interpolated configurations per segment (pybullet's controller sweeps the same joint-space
segment).  This is a proxy (no dynamics, box-shaped links), reported as such; it is not on the GPU hot path.
"""
from __future__ import annotations

import numpy as np

from . import franka


def _dh(a, d, alpha, q):
    cq, sq, ca, sa = np.cos(q), np.sin(q), np.cos(alpha), np.sin(alpha)
    return np.array([[cq, -sq, 0, a], [sq * ca, cq * ca, -sa, -sa * d], [sq * sa, cq * sa, ca, ca * d], [0, 0, 0, 1.0]])


def link_box_poses(q):
    """q (7,) -> list of 9 (R (3,3), center (3,)) world poses of the link boxes (float64 modified-DH chain)."""
    T = np.eye(4)
    frames = []
    for i in range(7):
        a, d, al = franka.DH_A_D_ALPHA[i]
        T = T @ _dh(a, d, al, q[i])
        frames.append(T.copy())
    sf = franka.static_frames().astype(np.float64)
    out = []
    for l in range(franka.N_LINKS):
        F = frames[franka.LINK_FRAME[l]]
        S = np.eye(4)
        S[:3, :] = sf[l]
        W = F @ S
        out.append((W[:3, :3], W[:3, 3]))
    return out


def quat_xyzw_to_matrix(q):
    x, y, z, w = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
        ]
    )


def obb_overlap(Ra, ca, ha, Rb, cb, hb, eps=1e-12) -> bool:
    """separating-axis test for two oriented boxes (R columns = axes, c centre, h half extents)."""
    R = Ra.T @ Rb
    t = Ra.T @ (cb - ca)
    A = np.abs(R) + eps
    for i in range(3):
        if abs(t[i]) > ha[i] + hb @ A[i]:
            return False
    for j in range(3):
        if abs(t @ R[:, j]) > ha @ A[:, j] + hb[j]:
            return False
    for i in range(3):
        for j in range(3):
            ra = ha[(i + 1) % 3] * A[(i + 2) % 3, j] + ha[(i + 2) % 3] * A[(i + 1) % 3, j]
            rb = hb[(j + 1) % 3] * A[i, (j + 2) % 3] + hb[(j + 2) % 3] * A[i, (j + 1) % 3]
            if abs(t[(i + 2) % 3] * R[(i + 1) % 3, j] - t[(i + 1) % 3] * R[(i + 2) % 3, j]) > ra + rb:
                return False
    return True


def configuration_in_collision(q, obstacle_config, link_mesh_extents=None) -> bool:
    he = franka.link_half_extents(link_mesh_extents).astype(np.float64)
    poses = link_box_poses(np.asarray(q, dtype=np.float64))
    for o in np.asarray(obstacle_config, dtype=np.float64):
        Ro, co, ho = quat_xyzw_to_matrix(o[3:7]), o[:3], o[7:10] / 2
        for l, (Rl, cl) in enumerate(poses):
            if obb_overlap(Rl, cl, he[l], Ro, co, ho):
                return True
    return False


def geometric_success(trajectory, obstacle_config, substeps: int = 4, link_mesh_extents=None) -> dict:
    """trajectory (7, N).  success = within joint limits and no link-box / obstacle-box intersection at any waypoint or
    interpolated configuration.  Returns dict(success, first_collision_waypoint, within_limits)."""
    tr = np.asarray(trajectory, dtype=np.float64)
    lo, hi = franka.joint_limits()
    within = bool(np.all(tr >= lo[:, None] - 1e-9) and np.all(tr <= hi[:, None] + 1e-9))
    n = tr.shape[1]
    first = -1
    for i in range(n):
        stops = [0.0] if i == n - 1 else [s / substeps for s in range(substeps)]
        for s in stops:
            q = tr[:, i] if s == 0.0 else (1 - s) * tr[:, i] + s * tr[:, i + 1]
            if configuration_in_collision(q, obstacle_config, link_mesh_extents):
                first = i
                break
        if first >= 0:
            break
    return dict(success=bool(within and first < 0), first_collision_waypoint=first, within_limits=within)


# the fixed flange / hand chain behind joint 7: rows 8-10 of the reference's modified-DH table [a, d, alpha, theta]
# (lib/guide.py:36-38), used only by get_end_effector_transform (lib/guide.py:100-116)
EE_STATIC_DH = ((0.0, 0.107, 0.0, 0.0), (0.0, 0.0, 0.0, -np.pi / 4), (0.0, 0.1034, 0.0, 0.0))


def end_effector_positions(trajectory):
    """(N, 3) end-effector positions as lib/metrics.py computes them: the translation of
    IntersectionVolumeGuide.get_end_effector_transform (lib/guide.py:100-116), i.e. ALL TEN modified-DH rows - the seven
    joints followed by the fixed rows d = 0.107, theta = -pi/4, d = 0.1034 (0.21 m beyond the joint-7 frame).
    float64 here, float32 in the reference: agrees to ~1e-7 m (pinned by tests/golden/g13_metrics.npz)."""
    tr = np.asarray(trajectory, dtype=np.float64)
    pts = []
    for i in range(tr.shape[1]):
        T = np.eye(4)
        for j in range(7):
            a, d, al = franka.DH_A_D_ALPHA[j]
            T = T @ _dh(a, d, al, tr[j, i])
        for a, d, al, th in EE_STATIC_DH:
            T = T @ _dh(a, d, al, th)
        pts.append(T[:3, 3])
    return np.array(pts)


def path_lengths(trajectory) -> dict:
    """MetricsCalculator.path_length_metric (lib/metrics.py:32-45): joint-space and end-effector path length."""
    tr = np.asarray(trajectory, dtype=np.float64)
    ee = end_effector_positions(tr)
    return dict(joint=float(np.sum(np.linalg.norm(np.diff(tr.T, 1, axis=0), axis=1))), end_effector=float(np.sum(np.linalg.norm(np.diff(ee, 1, axis=0), axis=1))))


def sparc(speed_profile, fs: float, padlevel: int = 4, fc: float = 10.0, amp_th: float = 0.05) -> float:
    """Spectral arc length smoothness (Balasubramanian et al. 2015): the value `MetricsCalculator.sparc`
    (lib/metrics.py:47-125, a restatement of mpinets/third_party/sparc.py) returns first.  More negative = less smooth;
    an all-zero profile returns 0 like the reference."""
    v = np.asarray(speed_profile, dtype=np.float64)
    if np.allclose(v, 0):
        return 0.0
    nfft = int(pow(2, np.ceil(np.log2(len(v))) + padlevel))
    f = np.arange(0, fs, fs / nfft)
    Mf = np.abs(np.fft.fft(v, nfft))
    Mf = Mf / Mf.max()
    sel = np.nonzero(f <= fc)[0]
    f_sel, Mf_sel = f[sel], Mf[sel]
    idx = np.nonzero(Mf_sel >= amp_th)[0]
    f_sel, Mf_sel = f_sel[idx[0] : idx[-1] + 1], Mf_sel[idx[0] : idx[-1] + 1]
    return float(-np.sum(np.sqrt((np.diff(f_sel) / (f_sel[-1] - f_sel[0])) ** 2 + np.diff(Mf_sel) ** 2)))


def smoothness_metric(trajectory, dt: float = 0.1) -> tuple:
    """MetricsCalculator.smoothness_metric (lib/metrics.py:11-30): (joint SPARC, end-effector SPARC) of the speed
    profiles ||diff / dt|| of the (7, N) joint trajectory and of its end-effector positions."""
    tr = np.asarray(trajectory, dtype=np.float64)
    js = np.linalg.norm(np.diff(tr.T, n=1, axis=0) / dt, axis=1)
    ee = end_effector_positions(tr)
    es = np.linalg.norm(np.diff(ee, n=1, axis=0) / dt, axis=1)
    return sparc(js, 1.0 / dt), sparc(es, 1.0 / dt)


def smoothness(trajectory, dt: float = 0.1) -> float:
    """joint-space SPARC of a (7, N) trajectory (first component of smoothness_metric)."""
    return smoothness_metric(trajectory, dt)[0]
