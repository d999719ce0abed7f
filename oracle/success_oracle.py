"""CPU oracle of the geometric success check — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/__init__.py).

What it stands for.  The reference scores a plan by executing it in pybullet: `RobotEnvironment.benchmark_trajectory`
(lib/environment.py:632-680) drives the arm through the waypoints under position control and `check_collisions`
(:591-608) asks the simulator for contact points between the manipulator and every spawned obstacle - cuboids
(`spawn_collision_cuboids` :230-247, half extents = dims / 2) and TRUE cylinders (`spawn_collision_cylinders`
:249-268, radius = config[7], height = config[8], axis = local z); success = no contact at any simulation step
(:672).  pybullet, its Franka meshes and the datasets are absent offline ("parity unpinned"), so the
criterion is restated geometrically and EXACTLY on the same primitives the guide uses for the robot: the 9 Franka
link boxes (lib/guide.py:243-342) in their float64 modified-DH poses against every obstacle - oriented boxes by the
15-axis separating-axis test, finite cylinders by an exact box / cylinder test - at every waypoint and at `substeps`
configurations interpolated in joint space per segment (the controller of `move_joints` :542-584 sweeps that
segment), plus the joint-limit test the reference only prints (:659-661).

This file is the checker of the HIP kernel `success_rows_kernel` (edmp_amd/csrc/success.hip): the kernel follows
the same arithmetic in float64, tests compare flags on random rows incl. constructed touching / just-separated
pairs.  Scalar functions (readable, used on small cases) and vectorised twins (used on hundreds of rows) are both
here and are tested against each other and - the cylinder test - against a numerical minimisation
(tests/test_success_oracle.py).
"""
from __future__ import annotations

import numpy as np

from edmp_amd import franka

SAT_EPS = 1e-12  # added to |R| in the box / box test: keeps near-parallel edge pairs from producing a null axis


# --------------------------------------------------------------------------------------------------------------
# robot: link boxes                                                            lib/guide.py:29-38, 74-98, 243-352
# --------------------------------------------------------------------------------------------------------------


def dh_f64() -> np.ndarray:
    """(7, 4) float64 [a, d, cos(alpha), sin(alpha)] of the modified-DH rows (lib/guide.py:29-36)."""
    t = np.zeros((7, 4))
    t[:, 0] = franka.DH_A_D_ALPHA[:, 0]
    t[:, 1] = franka.DH_A_D_ALPHA[:, 1]
    t[:, 2] = np.cos(franka.DH_A_D_ALPHA[:, 2])
    t[:, 3] = np.sin(franka.DH_A_D_ALPHA[:, 2])
    return t


def _dh(a, d, ca, sa, q):
    cq, sq = np.cos(q), np.sin(q)
    return np.array([[cq, -sq, 0, a], [sq * ca, cq * ca, -sa, -sa * d], [sq * sa, cq * sa, ca, ca * d], [0, 0, 0, 1.0]])


def link_box_poses(q):
    """q (7,) -> list of 9 (R (3,3), centre (3,)) world poses of the link boxes (float64 modified-DH chain,
    lib/guide.py:45-98; link frame = joint frame x static frame, :344-352)."""
    T = np.eye(4)
    frames = []
    tab = dh_f64()
    for i in range(7):
        T = T @ _dh(tab[i, 0], tab[i, 1], tab[i, 2], tab[i, 3], q[i])
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


def link_box_poses_batch(Q):
    """Q (M, 7) -> R (M, 9, 3, 3), c (M, 9, 3)."""
    Q = np.asarray(Q, dtype=np.float64)
    M = Q.shape[0]
    tab = dh_f64()
    T = np.broadcast_to(np.eye(4), (M, 4, 4)).copy()
    frames = []
    for i in range(7):
        a, d, ca, sa = tab[i]
        cq, sq = np.cos(Q[:, i]), np.sin(Q[:, i])
        D = np.zeros((M, 4, 4))
        D[:, 0, 0], D[:, 0, 1], D[:, 0, 3] = cq, -sq, a
        D[:, 1, 0], D[:, 1, 1], D[:, 1, 2], D[:, 1, 3] = sq * ca, cq * ca, -sa, -sa * d
        D[:, 2, 0], D[:, 2, 1], D[:, 2, 2], D[:, 2, 3] = sq * sa, cq * sa, ca, ca * d
        D[:, 3, 3] = 1.0
        T = T @ D
        frames.append(T)
    sf = franka.static_frames().astype(np.float64)
    R = np.zeros((M, 9, 3, 3))
    c = np.zeros((M, 9, 3))
    for l in range(franka.N_LINKS):
        S = np.eye(4)
        S[:3, :] = sf[l]
        W = frames[franka.LINK_FRAME[l]] @ S
        R[:, l] = W[:, :3, :3]
        c[:, l] = W[:, :3, 3]
    return R, c


def quat_xyzw_to_matrix(q):
    x, y, z, w = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
        ]
    )


# --------------------------------------------------------------------------------------------------------------
# box / box: separating-axis test (15 axes)                       stands for the simulator's contact query :591-608
# --------------------------------------------------------------------------------------------------------------


def obb_overlap(Ra, ca, ha, Rb, cb, hb, eps=SAT_EPS) -> bool:
    """two oriented boxes (R columns = axes, c centre, h half extents); touching counts as overlap."""
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


def obb_overlap_batch(Ra, ca, ha, Rb, cb, hb, eps=SAT_EPS):
    """vectorised twin: leading dimensions broadcast; returns a bool array."""
    R = np.einsum("...ki,...kj->...ij", Ra, Rb)
    t = np.einsum("...ki,...k->...i", Ra, cb - ca)
    A = np.abs(R) + eps
    ha = np.broadcast_to(ha, t.shape)
    hb = np.broadcast_to(hb, t.shape)
    sep = np.zeros(t.shape[:-1], dtype=bool)
    for i in range(3):
        sep |= np.abs(t[..., i]) > ha[..., i] + np.einsum("...j,...j->...", hb, A[..., i, :])
    for j in range(3):
        sep |= np.abs(np.einsum("...i,...i->...", t, R[..., :, j])) > np.einsum("...i,...i->...", ha, A[..., :, j]) + hb[..., j]
    for i in range(3):
        i1, i2 = (i + 1) % 3, (i + 2) % 3
        for j in range(3):
            j1, j2 = (j + 1) % 3, (j + 2) % 3
            ra = ha[..., i1] * A[..., i2, j] + ha[..., i2] * A[..., i1, j]
            rb = hb[..., j1] * A[..., i, j2] + hb[..., j2] * A[..., i, j1]
            sep |= np.abs(t[..., i2] * R[..., i1, j] - t[..., i1] * R[..., i2, j]) > ra + rb
    return ~sep


# --------------------------------------------------------------------------------------------------------------
# box / finite cylinder, exact                                                        lib/environment.py:249-268
# --------------------------------------------------------------------------------------------------------------
# In the cylinder's frame (axis = z, |z| <= H, x^2 + y^2 <= r^2) the box B clipped to the slab |z| <= H is a convex
# polytope P; the shapes meet iff min over P of x^2 + y^2 <= r^2.  That minimum is 0 iff the axis LINE pierces P (step 1);
# otherwise it is attained on the silhouette of P's projection onto the xy plane, which consists of projected edges of P:
# the 12 box edges clipped to the slab (step 2) and, on each cap plane z = +-H, the segments face-of-B x plane (step 3).
# Every edge of P is tested, so no adjacency / hull bookkeeping is needed: edges inside the silhouette only add
# candidates that are not smaller than the true minimum.


def _seg_dist2_origin(ax, ay, bx, by):
    dx, dy = bx - ax, by - ay
    dd = dx * dx + dy * dy
    s = 0.0 if dd == 0.0 else min(1.0, max(0.0, -(ax * dx + ay * dy) / dd))
    px, py = ax + s * dx, ay + s * dy
    return px * px + py * py


def _clip(lo, hi, p, d, h):
    """intersect the parameter interval [lo, hi] with |p + s d| <= h; returns (lo, hi), empty if lo > hi."""
    if d == 0.0:
        return (lo, hi) if abs(p) <= h else (1.0, 0.0)
    s0, s1 = (-h - p) / d, (h - p) / d
    if s0 > s1:
        s0, s1 = s1, s0
    return max(lo, s0), min(hi, s1)


def obb_cylinder_overlap(Rb, cb, hb, Rc, cc, radius, half_height) -> bool:
    """oriented box (Rb, cb, hb) against the finite cylinder of axis Rc[:, 2] through cc; touching counts as overlap."""
    R = Rc.T @ Rb          # box axes in the cylinder frame (columns)
    t = Rc.T @ (cb - cc)   # box centre in the cylinder frame
    H, r2 = half_height, radius * radius
    # step 0: the box's z range misses the slab
    if abs(t[2]) > H + hb[0] * abs(R[2, 0]) + hb[1] * abs(R[2, 1]) + hb[2] * abs(R[2, 2]):
        return False
    # step 1: the axis line (0, 0, z), |z| <= H, against the box: box coordinate i of the point is R[2, i] z - R[:, i] . t
    lo, hi = -H, H
    for i in range(3):
        lo, hi = _clip(lo, hi, -(R[0, i] * t[0] + R[1, i] * t[1] + R[2, i] * t[2]), R[2, i], hb[i])
    if lo <= hi:
        return True
    # step 2: the 12 box edges, clipped to the slab
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        for sj in (-1.0, 1.0):
            for sk in (-1.0, 1.0):
                p0 = t + sj * hb[j] * R[:, j] + sk * hb[k] * R[:, k] - hb[i] * R[:, i]
                d = 2.0 * hb[i] * R[:, i]
                s0, s1 = _clip(0.0, 1.0, p0[2], d[2], H)
                if s0 > s1:
                    continue
                if _seg_dist2_origin(p0[0] + s0 * d[0], p0[1] + s0 * d[1], p0[0] + s1 * d[0], p0[1] + s1 * d[1]) <= r2:
                    return True
    # step 3: on each cap plane, the segment cut out of each of the 6 box faces.  In face coordinates (a, b) along the face's
    # axes j, k the cut is the line nj a + nk b = e; it is parametrised by the coordinate with the SMALLER normal component
    # and solved for the other (division by the dominant component: bounded conditioning, a face nearly parallel to the cap
    # yields an empty clip instead of garbage)
    for cz in (H, -H):
        for i in range(3):
            j, k = (i + 1) % 3, (i + 2) % 3
            nj, nk = R[2, j], R[2, k]
            if abs(nk) > abs(nj):
                j, k, nj, nk = k, j, nk, nj
            if nj == 0.0:
                continue  # face parallel to the cap: its edges are box edges, step 2 has them
            for si in (-1.0, 1.0):
                f = t + si * hb[i] * R[:, i]
                p, d = (cz - f[2]) / nj, -nk / nj  # a(u) = p + d u,  b = u in [-hb[k], hb[k]]
                lo, hi = _clip(-hb[k], hb[k], p, d, hb[j])
                if lo > hi:
                    continue
                pa = f + (p + d * lo) * R[:, j] + lo * R[:, k]
                pb = f + (p + d * hi) * R[:, j] + hi * R[:, k]
                if _seg_dist2_origin(pa[0], pa[1], pb[0], pb[1]) <= r2:
                    return True
    return False


def _clip_b(lo, hi, p, d, h):
    with np.errstate(divide="ignore", invalid="ignore"):
        s0, s1 = (-h - p) / d, (h - p) / d
    a, b = np.minimum(s0, s1), np.maximum(s0, s1)
    z = d == 0.0
    inside = np.abs(p) <= h
    nlo = np.where(z, np.where(inside, lo, 1.0), np.maximum(lo, a))
    nhi = np.where(z, np.where(inside, hi, 0.0), np.minimum(hi, b))
    return nlo, nhi


def _seg_dist2_origin_b(ax, ay, bx, by):
    dx, dy = bx - ax, by - ay
    dd = dx * dx + dy * dy
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.where(dd == 0.0, 0.0, np.minimum(1.0, np.maximum(0.0, -(ax * dx + ay * dy) / dd)))
    px, py = ax + s * dx, ay + s * dy
    return px * px + py * py


def obb_cylinder_overlap_batch(Rb, cb, hb, Rc, cc, radius, half_height):
    """vectorised twin of obb_cylinder_overlap (leading dimensions broadcast)."""
    R = np.einsum("...ki,...kj->...ij", Rc, Rb)
    t = np.einsum("...ki,...k->...i", Rc, cb - cc)
    hb = np.broadcast_to(hb, t.shape)
    H = np.broadcast_to(np.asarray(half_height, dtype=np.float64), t.shape[:-1])
    r2 = np.broadcast_to(np.asarray(radius, dtype=np.float64) ** 2, t.shape[:-1])
    miss = np.abs(t[..., 2]) > H + hb[..., 0] * np.abs(R[..., 2, 0]) + hb[..., 1] * np.abs(R[..., 2, 1]) + hb[..., 2] * np.abs(R[..., 2, 2])
    lo, hi = -H, H
    for i in range(3):
        lo, hi = _clip_b(lo, hi, -(R[..., 0, i] * t[..., 0] + R[..., 1, i] * t[..., 1] + R[..., 2, i] * t[..., 2]), R[..., 2, i], hb[..., i])
    hit = lo <= hi
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        for sj in (-1.0, 1.0):
            for sk in (-1.0, 1.0):
                p0 = t + (sj * hb[..., j])[..., None] * R[..., :, j] + (sk * hb[..., k])[..., None] * R[..., :, k] - hb[..., i][..., None] * R[..., :, i]
                d = (2.0 * hb[..., i])[..., None] * R[..., :, i]
                s0, s1 = _clip_b(np.zeros_like(H), np.ones_like(H), p0[..., 2], d[..., 2], H)
                ok = s0 <= s1
                d2 = _seg_dist2_origin_b(p0[..., 0] + s0 * d[..., 0], p0[..., 1] + s0 * d[..., 1], p0[..., 0] + s1 * d[..., 0], p0[..., 1] + s1 * d[..., 1])
                hit |= ok & (d2 <= r2)
    for sign in (1.0, -1.0):
        cz = sign * H
        for i in range(3):
            j, k = (i + 1) % 3, (i + 2) % 3
            swap = np.abs(R[..., 2, k]) > np.abs(R[..., 2, j])
            nj = np.where(swap, R[..., 2, k], R[..., 2, j])
            nk = np.where(swap, R[..., 2, j], R[..., 2, k])
            Rj = np.where(swap[..., None], R[..., :, k], R[..., :, j])
            Rk = np.where(swap[..., None], R[..., :, j], R[..., :, k])
            hj = np.where(swap, hb[..., k], hb[..., j])
            hk = np.where(swap, hb[..., j], hb[..., k])
            par = nj == 0.0
            for si in (-1.0, 1.0):
                f = t + (si * hb[..., i])[..., None] * R[..., :, i]
                with np.errstate(divide="ignore", invalid="ignore"):
                    p, d = (cz - f[..., 2]) / nj, -nk / nj
                p, d = np.where(par, 0.0, p), np.where(par, 0.0, d)
                lo, hi = _clip_b(-hk, hk, p, d, hj)
                ok = (lo <= hi) & ~par
                lo_, hi_, p_ = np.where(ok, lo, 0.0), np.where(ok, hi, 0.0), np.where(ok, p, 0.0)
                pa = f + (p_ + d * lo_)[..., None] * Rj + lo_[..., None] * Rk
                pb = f + (p_ + d * hi_)[..., None] * Rj + hi_[..., None] * Rk
                hit |= ok & (_seg_dist2_origin_b(pa[..., 0], pa[..., 1], pb[..., 0], pb[..., 1]) <= r2)
    return hit & ~miss


# --------------------------------------------------------------------------------------------------------------
# trajectories                                                                        lib/environment.py:632-680
# --------------------------------------------------------------------------------------------------------------


def obstacle_shapes(obstacle_config, kinds=None):
    """-> R (no,3,3), c (no,3), half (no,3), kinds (no,) int (0 cuboid, 1 cylinder with dims (r, r, h): radius = dims[0],
    half height = dims[2] / 2 - the (r, r, h) row the reference's loader builds, load_test_dataset.py:136-139)."""
    oc = np.asarray(obstacle_config, dtype=np.float64)
    k = np.zeros(oc.shape[0], dtype=np.int32) if kinds is None else np.asarray(kinds, dtype=np.int32)
    R = np.stack([quat_xyzw_to_matrix(o[3:7]) for o in oc])
    return R, oc[:, :3].copy(), oc[:, 7:10] / 2, k


def configuration_in_collision(q, obstacle_config, kinds=None, link_mesh_extents=None) -> bool:
    he = franka.link_half_extents(link_mesh_extents).astype(np.float64)
    poses = link_box_poses(np.asarray(q, dtype=np.float64))
    Ro, co, ho, kd = obstacle_shapes(obstacle_config, kinds)
    for o in range(len(co)):
        for l, (Rl, cl) in enumerate(poses):
            if kd[o] == 1:
                if obb_cylinder_overlap(Rl, cl, he[l], Ro[o], co[o], 2 * ho[o, 0], ho[o, 2]):
                    return True
            elif obb_overlap(Rl, cl, he[l], Ro[o], co[o], ho[o]):
                return True
    return False


def interpolated_configurations(tr, substeps):
    """(7, N) -> ((N-1)*substeps + 1, 7): waypoint i, then (1 - s) q_i + s q_{i+1} for s = 1/S .. (S-1)/S; the last
    waypoint once."""
    tr = np.asarray(tr, dtype=np.float64)
    n = tr.shape[1]
    out = []
    for i in range(n):
        stops = [0.0] if i == n - 1 else [s / substeps for s in range(substeps)]
        for s in stops:
            out.append(tr[:, i] if s == 0.0 else (1 - s) * tr[:, i] + s * tr[:, i + 1])
    return np.array(out)


def geometric_success(trajectory, obstacle_config, substeps: int = 4, kinds=None, link_mesh_extents=None) -> dict:
    """scalar reference of ONE trajectory (7, N): dict(success, first_collision_waypoint, within_limits)."""
    tr = np.asarray(trajectory, dtype=np.float64)
    lo, hi = franka.joint_limits()
    within = bool(np.all(tr >= lo[:, None] - 1e-9) and np.all(tr <= hi[:, None] + 1e-9))
    first = -1
    for c, q in enumerate(interpolated_configurations(tr, substeps)):
        if configuration_in_collision(q, obstacle_config, kinds, link_mesh_extents):
            first = c // substeps
            break
    return dict(success=bool(within and first < 0), first_collision_waypoint=first, within_limits=within)


def success_rows(X, obstacle_config, substeps: int = 4, kinds=None, link_mesh_extents=None) -> dict:
    """vectorised over rows: X (B, 7, N) -> dict(ok (B,) bool, first (B,) int32 (-1 = none), within (B,) bool)."""
    X = np.asarray(X, dtype=np.float64)
    B, _, N = X.shape
    lo, hi = franka.joint_limits()
    within = np.all(X >= lo[None, :, None] - 1e-9, axis=(1, 2)) & np.all(X <= hi[None, :, None] + 1e-9, axis=(1, 2))
    S = int(substeps)
    nc = (N - 1) * S + 1
    Q = np.zeros((B, nc, 7))
    for i in range(N):
        for s in range(S if i < N - 1 else 1):
            f = s / S
            Q[:, i * S + s] = X[:, :, i] if s == 0 else (1 - f) * X[:, :, i] + f * X[:, :, i + 1]
    he = franka.link_half_extents(link_mesh_extents).astype(np.float64)
    Rl, cl = link_box_poses_batch(Q.reshape(-1, 7))  # (M, 9, 3, 3), (M, 9, 3)
    Ro, co, ho, kd = obstacle_shapes(obstacle_config, kinds)
    hit = np.zeros(B * nc, dtype=bool)
    for o in range(len(co)):
        if kd[o] == 1:
            hit |= obb_cylinder_overlap_batch(Rl, cl, he[None], Ro[o], co[o], 2 * ho[o, 0], ho[o, 2]).any(axis=1)
        else:
            hit |= obb_overlap_batch(Rl, cl, he[None], Ro[o], co[o], ho[o]).any(axis=1)
    hit = hit.reshape(B, nc)
    anyhit = hit.any(axis=1)
    first = np.where(anyhit, hit.argmax(axis=1) // S, -1).astype(np.int32)
    return dict(ok=within & ~anyhit, first=first, within=within)
