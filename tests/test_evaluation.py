"""CPU: host-side evaluation utilities (metrics, scene files) — SURVEY.md §8f rows 1 and 3.  The success check runs on the
GPU (tests/test_gpu_success.py); its CPU checker is tested in tests/test_success_oracle.py."""
import json

import numpy as np
import pytest

from edmp_amd import evaluation as EV
from edmp_amd import franka, scenes


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


def test_synthetic_dataset_with_cylinders_follows_the_loader_contract():
    ds = scenes.SyntheticDataset(scene_types=("stress",), n_obstacles=6, n_cylinders=2)
    oc, cub, cyl, nb, ncyl, start, iks = ds.fetch_data(0, "stress")
    assert (nb, ncyl) == (4, 2) and cub.shape == (4, 10) and cyl.shape == (2, 9) and oc.shape == (6, 10)
    assert np.array_equal(oc[:4], cub) and np.array_equal(oc[4:, 7], oc[4:, 8])  # cylinders enter as (r, r, h) boxes, after the cuboids
    assert np.array_equal(cyl[:, 7], oc[4:, 7]) and np.array_equal(cyl[:, 8], oc[4:, 9])
