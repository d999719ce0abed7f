"""CPU: host-side evaluation utilities (success proxy, metrics, scene files) — SURVEY.md §8f rows 1 and 3."""
import json

import numpy as np
import pytest

from edmp_amd import evaluation as EV
from edmp_amd import franka, scenes


def test_host_fk_matches_oracle_fk():
    import torch

    from oracle import edmp_oracle as O

    rs = np.random.RandomState(0)
    lo, hi = franka.joint_limits()
    q = rs.uniform(lo, hi)
    lt = O.get_link_transform(torch.tensor(q[None, None, :], dtype=torch.float32))[0, 0].numpy()
    for l, (R, c) in enumerate(EV.link_box_poses(q)):
        assert np.allclose(R, lt[l, :3, :3], atol=2e-6) and np.allclose(c, lt[l, :3, 3], atol=2e-6)


def test_obb_overlap_basics():
    I = np.eye(3)
    h = np.array([0.5, 0.5, 0.5])
    assert EV.obb_overlap(I, np.zeros(3), h, I, np.array([0.9, 0, 0]), h)
    assert not EV.obb_overlap(I, np.zeros(3), h, I, np.array([1.1, 0, 0]), h)
    # rotated 45 deg about z: corner reaches sqrt(2)/2
    Rz = EV.quat_xyzw_to_matrix([0, 0, np.sin(np.pi / 8), np.cos(np.pi / 8)])
    assert EV.obb_overlap(I, np.zeros(3), h, Rz, np.array([1.15, 0, 0]), h)
    assert not EV.obb_overlap(I, np.zeros(3), h, Rz, np.array([1.25, 0, 0]), h)
    # an edge-edge separating axis case (AABB of the rotated box overlaps, the boxes do not)
    Rx = EV.quat_xyzw_to_matrix([np.sin(np.pi / 8), 0, 0, np.cos(np.pi / 8)])
    assert not EV.obb_overlap(Rz, np.zeros(3), h, Rx @ Rz, np.array([1.05, 1.05, 0.0]), h)


def test_geometric_success_and_exactness_vs_aabb_guide():
    lo, hi = franka.joint_limits()
    start = scenes.DEFAULT_START
    traj = np.tile(start[:, None], (1, 50))
    far = np.array([[5.0, 0, 0, 0, 0, 0, 1, 0.2, 0.2, 0.2]])
    assert EV.geometric_success(traj, far)["success"]
    # put a box exactly on link 5's centre -> collision
    R, c = EV.link_box_poses(start)[4]
    hit = np.array([[*c, 0, 0, 0, 1, 0.1, 0.1, 0.1]])
    r = EV.geometric_success(traj, hit)
    assert not r["success"] and r["first_collision_waypoint"] == 0
    out = traj.copy()
    out[3, 7] = 0.3  # joint 4 upper limit is -4 deg
    assert not EV.geometric_success(out, far)["success"] and not EV.geometric_success(out, far)["within_limits"]
    # collision only between waypoints is caught by the interpolation
    q0, q1 = start.copy(), start.copy()
    q1[0] += 1.2
    seg = np.concatenate([np.tile(q0[:, None], (1, 25)), np.tile(q1[:, None], (1, 25))], axis=1)
    qm = 0.5 * (q0 + q1)
    Rm, cm = EV.link_box_poses(qm)[6]
    mid = np.array([[*cm, 0, 0, 0, 1, 0.02, 0.02, 0.02]])
    assert not EV.configuration_in_collision(q0, mid) and not EV.configuration_in_collision(q1, mid)
    assert not EV.geometric_success(seg, mid, substeps=8)["success"]


def test_metrics():
    t = np.linspace(0, 1, 50)
    lo, hi = franka.joint_limits()
    a, b = scenes.DEFAULT_START, scenes.DEFAULT_GOAL
    straight = a[:, None] * (1 - t) + b[:, None] * t
    pl = EV.path_lengths(straight)
    assert abs(pl["joint"] - np.linalg.norm(b - a)) < 1e-9 and pl["end_effector"] > 0
    rs = np.random.RandomState(0)
    jerky = straight + 0.05 * rs.standard_normal(straight.shape)
    smooth_bell = a[:, None] + (b - a)[:, None] * (10 * t**3 - 15 * t**4 + 6 * t**5)
    assert EV.smoothness(smooth_bell) > EV.smoothness(jerky)  # SPARC: closer to 0 = smoother
    assert EV.path_lengths(jerky)["joint"] > pl["joint"]


def test_problem_file_round_trip(tmp_path):
    oc = scenes.random_scene(3, 5)
    s, g = scenes.random_start_goal(3)
    p = tmp_path / "scene.json"
    scenes.save_problem_file(str(p), oc, s, np.stack([g, s]))
    oc2, s2, g2 = scenes.load_problem_file(str(p))
    assert np.allclose(oc2, oc) and np.allclose(s2, s) and g2.shape == (2, 7)
    # scalar-first -> scalar-last roll and cylinder -> (r, r, h) box (quirk Q9)
    prob = {"cuboids": [{"center": [0.5, 0, 0.2], "quaternion_wxyz": [1, 0, 0, 0], "dims": [0.1, 0.2, 0.3]}],
            "cylinders": [{"center": [0.4, 0.1, 0.3], "quaternion_wxyz": [0.5, 0.5, 0.5, 0.5], "radius": 0.07, "height": 0.4}],
            "start": s.tolist(), "goals": [g.tolist()]}
    oc3, _, _ = scenes.problem_to_arrays(prob)
    assert oc3[0, 3:7].tolist() == [0, 0, 0, 1] and oc3[1, 7:].tolist() == [0.07, 0.07, 0.4] and oc3[1, 3:7].tolist() == [0.5, 0.5, 0.5, 0.5]
    with pytest.raises(ValueError):
        scenes.problem_to_arrays({"cuboids": [], "start": s.tolist(), "goals": [g.tolist()]})


def test_metrics_against_the_reference(golden):
    """G13 (oracle/gen_golden_metrics.py): path length and SPARC of six trajectories computed by the UNMODIFIED
    lib/metrics.py MetricsCalculator (end effector through the reference's 10-row DH chain, f32) and by the vendored
    mpinets/third_party/sparc.py."""
    g = golden("g13_metrics")
    dt = float(g["dt"])
    for i, tr in enumerate(g["trajectories"]):
        ee = EV.end_effector_positions(tr)
        assert np.abs(ee - g["ee_positions"][i]).max() <= 2e-6, i  # f64 here vs the reference's f32 chain
        pl = EV.path_lengths(tr)
        assert abs(pl["joint"] - g["joint_path_length"][i]) <= 1e-9 * max(1.0, g["joint_path_length"][i])
        assert abs(pl["end_effector"] - g["ee_path_length"][i]) <= 2e-5 * max(1.0, g["ee_path_length"][i])
        sj, se = EV.smoothness_metric(tr, dt)
        assert abs(sj - g["joint_sparc"][i]) <= 1e-9 and abs(sj - g["third_party_joint_sparc"][i]) <= 1e-9, (i, sj, g["joint_sparc"][i])
        assert abs(se - g["ee_sparc"][i]) <= 5e-4 * max(1.0, abs(g["ee_sparc"][i])), (i, se, g["ee_sparc"][i])  # f32 positions feed an FFT threshold
    assert EV.smoothness_metric(g["trajectories"][4], dt) == (0.0, 0.0)  # constant trajectory: the reference returns 0
