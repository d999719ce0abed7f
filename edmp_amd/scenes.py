"""Synthetic planning problems (the MPiNets problem pickles the reference loads are not redistributable/offline).

Output contract = what ``TestDataset.fetch_data`` hands the guide (datasets/load_test_dataset.py:76-189):
``obstacle_config`` (no, 10) float64 rows ``[cx, cy, cz, qx, qy, qz, qw, sx, sy, sz]`` (quaternion scalar-last,
full extents; cylinders enter as boxes with extents (r, r, h), load_test_dataset.py:136-139), a start
configuration (7,) and IK goal candidates (n, 7).  Generators follow SURVEY.md §8(d).
"""
from __future__ import annotations

import numpy as np

from .franka import joint_limits

DEFAULT_START = np.array([0.0, -0.5, 0.0, -2.0, 0.0, 1.6, 0.8])
DEFAULT_GOAL = np.array([0.6, 0.3, -0.4, -1.5, 0.2, 1.9, 0.2])


def random_scene(seed: int, n_obstacles: int = 8, yaw_only: bool = False) -> np.ndarray:
    rs = np.random.RandomState(seed)
    c = np.stack(
        [rs.uniform(0.2, 0.8, n_obstacles), rs.uniform(-0.6, 0.6, n_obstacles), rs.uniform(0.0, 0.8, n_obstacles)], axis=1
    )
    if yaw_only:
        yaw = rs.uniform(-np.pi, np.pi, n_obstacles)
        q = np.stack([np.zeros(n_obstacles), np.zeros(n_obstacles), np.sin(yaw / 2), np.cos(yaw / 2)], axis=1)
    else:
        q = rs.standard_normal((n_obstacles, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    d = rs.uniform(0.05, 0.4, (n_obstacles, 3))
    return np.concatenate([c, q, d], axis=1)


def random_start_goal(seed: int):
    lo, hi = joint_limits()
    rs = np.random.RandomState(seed + 1)
    return rs.uniform(lo, hi), rs.uniform(lo, hi)


def cylinder_as_box(center, quat_xyzw, radius, height) -> np.ndarray:
    """one obstacle_config row for a cylinder, the way the reference feeds it to the guide (quirk Q9)."""
    return np.concatenate([np.asarray(center, float), np.asarray(quat_xyzw, float), [radius, radius, height]])


class SyntheticDataset:
    """Stand-in for ``TestDataset`` (datasets/load_test_dataset.py:15-189) with the same ``fetch_data`` output
    contract: (obstacle_config, cuboid_config, cylinder_config, num_cuboids, num_cylinders, start_joints,
    all_ik_goals).  scene_type 'tabletop' = yaw-only cuboids, anything else = free orientations.  IK goals are seeded
    in-limit configurations (robofin/ikfast are not available)."""

    def __init__(self, dataset_type="synthetic", d_path=None, scene_types=("stress",), num_scenes_per_type=1, n_obstacles=8, n_ik=100):
        self.dataset_type = dataset_type
        self.n_obstacles, self.n_ik = int(n_obstacles), int(n_ik)
        n = 1 if num_scenes_per_type is None or num_scenes_per_type < 0 else int(num_scenes_per_type)
        self.data_nums = {st: n for st in scene_types}

    def fetch_data(self, scene_num, scene_type="stress"):
        seed = 1000 * (sum(map(ord, scene_type)) % 97) + int(scene_num)
        oc = random_scene(seed, self.n_obstacles, yaw_only=(scene_type == "tabletop"))
        start, _ = random_start_goal(seed)
        lo, hi = joint_limits()
        iks = np.random.RandomState(seed + 2).uniform(lo, hi, (self.n_ik, 7))
        return oc, oc.copy(), np.zeros((0, 10)), oc.shape[0], 0, start, iks
