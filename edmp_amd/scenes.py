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
