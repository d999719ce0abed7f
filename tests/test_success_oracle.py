"""CPU: the success-check oracle (oracle/success_oracle.py) against itself (scalar vs vectorised) and - the exact
box / cylinder test - against a numerical minimisation.  SURVEY.md §8f row 3."""
import numpy as np
import pytest

from edmp_amd import franka, scenes
from oracle import success_oracle as SO


def _rand_rot(rs):
    q = rs.standard_normal(4)
    return SO.quat_xyzw_to_matrix(q / np.linalg.norm(q))


def _axis_distance_numeric(Rb, cb, hb, Rc, cc, H):
    """min over (box ∩ slab) of the distance to the cylinder axis, by SLSQP in box coordinates; None if the set is empty."""
    from scipy.optimize import minimize

    R = Rc.T @ Rb
    t = Rc.T @ (cb - cc)
    if abs(t[2]) > H + hb @ np.abs(R[2]):
        return None
    f = lambda u: float(np.sum((t + R @ u)[:2] ** 2))  # noqa: E731
    g = lambda u: 2 * R[:2].T @ (t + R @ u)[:2]  # noqa: E731
    cons = [{"type": "ineq", "fun": lambda u: H - (t + R @ u)[2], "jac": lambda u: -R[2]},
            {"type": "ineq", "fun": lambda u: H + (t + R @ u)[2], "jac": lambda u: R[2]}]
    best = None
    for u0 in (np.zeros(3), hb * 0.9, -hb * 0.9):
        r = minimize(f, u0, jac=g, bounds=[(-h, h) for h in hb], constraints=cons, method="SLSQP", options={"ftol": 1e-15, "maxiter": 300})
        if r.success and (best is None or r.fun < best):
            best = r.fun
    return None if best is None else float(np.sqrt(max(best, 0.0)))


def test_box_cylinder_against_numerical_minimisation():
    rs = np.random.RandomState(5)
    n_dec = n_hit = 0
    for _ in range(400):
        Rb, Rc = _rand_rot(rs), _rand_rot(rs)
        hb = rs.uniform(0.03, 0.3, 3)
        r, H = rs.uniform(0.03, 0.3), rs.uniform(0.03, 0.4)
        cc = rs.uniform(-0.2, 0.2, 3)
        cb = cc + rs.standard_normal(3) * 0.35
        d = _axis_distance_numeric(Rb, cb, hb, Rc, cc, H)
        got = SO.obb_cylinder_overlap(Rb, cb, hb, Rc, cc, r, H)
        if d is None:
            assert not got
            continue
        if abs(d - r) < 1e-5:
            continue  # the optimiser is not sharper than this
        n_dec += 1
        n_hit += got
        assert got == (d < r), (d, r, got)
    assert n_dec > 250 and 0.2 < n_hit / n_dec < 0.8


def test_box_cylinder_constructed_cases():
    I = np.eye(3)
    h = np.array([0.1, 0.1, 0.1])
    z0 = np.zeros(3)
    # side contact along x: box face at distance r (touching counts), just separated, just inside
    assert SO.obb_cylinder_overlap(I, np.array([0.3, 0, 0]), h, I, z0, 0.2, 0.5)
    assert not SO.obb_cylinder_overlap(I, np.array([0.3 + 1e-9, 0, 0]), h, I, z0, 0.2, 0.5)
    # the bounding BOX of the cylinder (r, r, h as full extents 2r) would hit at the corner, the cylinder does not
    c = np.array([0.2 + 0.1 - 0.02, 0.2 + 0.1 - 0.02, 0.0])
    assert SO.obb_overlap(I, c, h, I, z0, np.array([0.2, 0.2, 0.5]))
    assert not SO.obb_cylinder_overlap(I, c, h, I, z0, 0.2, 0.5)
    # cap contact from above, and a box hovering over the cap rim only (edge of the cap polygon decides)
    assert SO.obb_cylinder_overlap(I, np.array([0, 0, 0.6]), h, I, z0, 0.2, 0.5)
    assert not SO.obb_cylinder_overlap(I, np.array([0, 0, 0.6 + 1e-9]), h, I, z0, 0.2, 0.5)
    Rt = SO.quat_xyzw_to_matrix([np.sin(0.3), 0, 0, np.cos(0.3)])
    assert SO.obb_cylinder_overlap(Rt, np.array([0.0, 0.25, 0.55]), h, I, z0, 0.2, 0.5)
    assert not SO.obb_cylinder_overlap(Rt, np.array([0.0, 0.36, 0.62]), h, I, z0, 0.2, 0.5)
    # axis pierces a large thin plate
    assert SO.obb_cylinder_overlap(Rt, np.array([0.0, 0.0, 0.1]), np.array([1.0, 1.0, 0.001]), I, z0, 0.01, 0.5)


def test_vectorised_twins_agree_with_the_scalar_functions():
    rs = np.random.RandomState(11)
    n = 300
    Ra = np.stack([_rand_rot(rs) for _ in range(n)])
    Rb = np.stack([_rand_rot(rs) for _ in range(n)])
    ca = rs.uniform(-0.2, 0.2, (n, 3))
    cb = ca + rs.standard_normal((n, 3)) * 0.3
    ha = rs.uniform(0.03, 0.3, (n, 3))
    hb = rs.uniform(0.03, 0.3, (n, 3))
    r, H = rs.uniform(0.03, 0.3, n), rs.uniform(0.03, 0.4, n)
    # axis-aligned / parallel cases exercise the d == 0 branches
    Ra[:20] = np.eye(3)
    Rb[:20] = np.eye(3)
    Rb[20:30] = Ra[20:30]
    bb = SO.obb_overlap_batch(Ra, ca, ha, Rb, cb, hb)
    bc = SO.obb_cylinder_overlap_batch(Ra, ca, ha, Rb, cb, r, H)
    for i in range(n):
        assert bb[i] == SO.obb_overlap(Ra[i], ca[i], ha[i], Rb[i], cb[i], hb[i]), i
        assert bc[i] == SO.obb_cylinder_overlap(Ra[i], ca[i], ha[i], Rb[i], cb[i], r[i], H[i]), i
    assert 0.1 < bb.mean() < 0.9 and 0.1 < bc.mean() < 0.9


def test_fk_batch_and_oracle_fk():
    import torch

    from oracle import edmp_oracle as O

    rs = np.random.RandomState(0)
    lo, hi = franka.joint_limits()
    Q = rs.uniform(lo, hi, (5, 7))
    Rb, cb = SO.link_box_poses_batch(Q)
    for m in range(5):
        lt = O.get_link_transform(torch.tensor(Q[m][None, None, :], dtype=torch.float32))[0, 0].numpy()
        for l, (R, c) in enumerate(SO.link_box_poses(Q[m])):
            assert np.allclose(R, lt[l, :3, :3], atol=2e-6) and np.allclose(c, lt[l, :3, 3], atol=2e-6)
            assert np.abs(R - Rb[m, l]).max() < 1e-14 and np.abs(c - cb[m, l]).max() < 1e-14


def test_rows_against_the_scalar_checker():
    rs = np.random.RandomState(3)
    lo, hi = franka.joint_limits()
    oc = scenes.random_scene(21, 6)
    kinds = np.array([0, 1, 0, 1, 0, 0], dtype=np.int32)
    oc[kinds == 1, 8] = oc[kinds == 1, 7]  # (r, r, h)
    B, N = 12, 10
    a, b = rs.uniform(lo, hi, (B, 7)), rs.uniform(lo, hi, (B, 7))
    t = np.linspace(0, 1, N)
    X = a[:, :, None] * (1 - t) + b[:, :, None] * t
    X[3, 3, 4] = 0.5  # outside the limits
    res = SO.success_rows(X, oc, substeps=3, kinds=kinds)
    for r in range(B):
        s = SO.geometric_success(X[r], oc, substeps=3, kinds=kinds)
        assert (bool(res["ok"][r]), int(res["first"][r]), bool(res["within"][r])) == (s["success"], s["first_collision_waypoint"], s["within_limits"]), r
    assert not res["within"][3] and res["ok"].sum() < B


def test_geometric_success_and_exactness_vs_aabb_guide():
    lo, hi = franka.joint_limits()
    start = scenes.DEFAULT_START
    traj = np.tile(start[:, None], (1, 50))
    far = np.array([[5.0, 0, 0, 0, 0, 0, 1, 0.2, 0.2, 0.2]])
    assert SO.geometric_success(traj, far)["success"]
    # put a box exactly on link 5's centre -> collision
    R, c = SO.link_box_poses(start)[4]
    hit = np.array([[*c, 0, 0, 0, 1, 0.1, 0.1, 0.1]])
    r = SO.geometric_success(traj, hit)
    assert not r["success"] and r["first_collision_waypoint"] == 0
    out = traj.copy()
    out[3, 7] = 0.3  # joint 4 upper limit is -4 deg
    assert not SO.geometric_success(out, far)["success"] and not SO.geometric_success(out, far)["within_limits"]
    # collision only between waypoints is caught by the interpolation
    q0, q1 = start.copy(), start.copy()
    q1[0] += 1.2
    seg = np.concatenate([np.tile(q0[:, None], (1, 25)), np.tile(q1[:, None], (1, 25))], axis=1)
    qm = 0.5 * (q0 + q1)
    Rm, cm = SO.link_box_poses(qm)[6]
    mid = np.array([[*cm, 0, 0, 0, 1, 0.02, 0.02, 0.02]])
    assert not SO.configuration_in_collision(q0, mid) and not SO.configuration_in_collision(q1, mid)
    assert not SO.geometric_success(seg, mid, substeps=8)["success"]
