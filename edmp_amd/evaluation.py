"""Result evaluation: plan success over the whole batch (on the GPU) and trajectory metrics (host, like the reference's).

Success.  The reference scores a plan by executing it in pybullet (`RobotEnvironment.benchmark_trajectory`,
lib/environment.py:632-680: position control through the waypoints, contact query `check_collisions` :591-608 against
spawned cuboids AND true cylinders :230-268) and tallies `t_success` (infer_serial.py:94-99,165-168).  pybullet is not
available offline, so the criterion is restated geometrically — EXACT oriented-box / finite-cylinder tests instead of the
guide's conservative world-AABB overlap — and evaluated for every row of the batch by `edmp_success_rows_dev`
(csrc/success.hip) through `IntersectionVolumeGuide.success_rows`.  It is a stand-in (no dynamics, box-shaped links),
reported as such.  There is no host fallback: the checker of the kernel lives in oracle/success_oracle.py (tests only).

Metrics.  lib/metrics.py:11-125 (`MetricsCalculator`, host NumPy / torch-CPU in the reference too, never called by its
driver): path length and SPARC smoothness, pinned to the reference by tests/golden/g13_metrics.npz.
"""
from __future__ import annotations

import numpy as np

from . import franka


def _dh(a, d, alpha, q):
    cq, sq, ca, sa = np.cos(q), np.sin(q), np.cos(alpha), np.sin(alpha)
    return np.array([[cq, -sq, 0, a], [sq * ca, cq * ca, -sa, -sa * d], [sq * sa, cq * sa, ca, ca * d], [0, 0, 0, 1.0]])


def geometric_success(trajectory, guide, substeps: int = 4) -> dict:
    """ONE trajectory (7, N) against the scene of `guide` (an IntersectionVolumeGuide): dict(success, collision_free,
    first_collision_waypoint, within_limits).  `collision_free` is the reference's success flag (lib/environment.py:672: contact
    only; leaving the limits merely prints, :659-661), `success` additionally requires `within_limits`.  Runs on the GPU (one-row
    batch of guide.success_rows)."""
    tr = np.asarray(trajectory, dtype=np.float64)
    if tr.ndim != 2 or tr.shape[0] != 7:
        raise ValueError(f"trajectory must be (7, N), got {tr.shape}")
    r = guide.success_rows(tr[None], substeps=substeps)
    return dict(success=bool(r["ok"][0]), collision_free=bool(r["collision_free"][0]), first_collision_waypoint=int(r["first"][0]), within_limits=bool(r["within"][0]))


def success_rate(trajectories, guide, substeps: int = 4) -> dict:
    """every row of a batch (B, 7, N): dict(rows_ok, rows, rate, ok (B,), first (B,), within (B,)) - the batch form of the
    reference's running tally `t_success / i` (infer_serial.py:99)."""
    r = guide.success_rows(trajectories, substeps=substeps)
    r["rate"] = r["rows_ok"] / max(r["rows"], 1)
    r["collision_free_rate"] = r["rows_collision_free"] / max(r["rows"], 1)  # the reference's criterion (lib/environment.py:672)
    return r


# the fixed flange / hand chain behind joint 7: rows 8-10 of the reference's modified-DH table [a, d, alpha, theta]
# (lib/guide.py:36-38), used only by get_end_effector_transform (lib/guide.py:100-116)
EE_STATIC_DH = ((0.0, 0.107, 0.0, 0.0), (0.0, 0.0, 0.0, -np.pi / 4), (0.0, 0.1034, 0.0, 0.0))


def end_effector_positions(trajectory):
    """(N, 3) end-effector positions as lib/metrics.py computes them: the translation of
    IntersectionVolumeGuide.get_end_effector_transform (lib/guide.py:100-116), i.e. ALL TEN modified-DH rows - the seven
    joints followed by the fixed rows d = 0.107, theta = -pi/4, d = 0.1034 (0.21 m beyond the joint-7 frame).
    float64 here, float32 in the reference: agrees to ~1e-7 m (pinned by tests/golden/g13_metrics.npz)."""
    tr = np.asarray(trajectory, dtype=np.float64)
    n = tr.shape[1]

    def dh_stack(a, d, alpha, q):  # (N, 4, 4): _dh for every waypoint at once (a per-waypoint Python loop cost 2 ms per call)
        cq, sq, ca, sa = np.cos(q), np.sin(q), np.cos(alpha), np.sin(alpha)
        D = np.zeros((n, 4, 4))
        D[:, 0, 0], D[:, 0, 1], D[:, 0, 3] = cq, -sq, a
        D[:, 1, 0], D[:, 1, 1], D[:, 1, 2], D[:, 1, 3] = sq * ca, cq * ca, -sa, -sa * d
        D[:, 2, 0], D[:, 2, 1], D[:, 2, 2], D[:, 2, 3] = sq * sa, cq * sa, ca, ca * d
        D[:, 3, 3] = 1.0
        return D

    T = np.broadcast_to(np.eye(4), (n, 4, 4))
    for j in range(7):
        a, d, al = franka.DH_A_D_ALPHA[j]
        T = T @ dh_stack(a, d, al, tr[j])
    for a, d, al, th in EE_STATIC_DH:
        T = T @ _dh(a, d, al, th)
    return np.ascontiguousarray(T[:, :3, 3])


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
