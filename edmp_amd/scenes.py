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

    def __init__(self, dataset_type="synthetic", d_path=None, scene_types=("stress",), num_scenes_per_type=1, n_obstacles=8, n_ik=100, n_cylinders=0):
        self.dataset_type = dataset_type
        self.n_obstacles, self.n_ik, self.n_cylinders = int(n_obstacles), int(n_ik), int(n_cylinders)
        if not 0 <= self.n_cylinders <= self.n_obstacles:
            raise ValueError("n_cylinders must lie in [0, n_obstacles]")
        n = 1 if num_scenes_per_type is None or num_scenes_per_type < 0 else int(num_scenes_per_type)
        self.data_nums = {st: n for st in scene_types}

    def fetch_data(self, scene_num, scene_type="stress"):
        seed = 1000 * (sum(map(ord, scene_type)) % 97) + int(scene_num)
        oc = random_scene(seed, self.n_obstacles, yaw_only=(scene_type == "tabletop"))
        start, _ = random_start_goal(seed)
        lo, hi = joint_limits()
        iks = np.random.RandomState(seed + 2).uniform(lo, hi, (self.n_ik, 7))
        # like the reference's loader: cuboids first, then cylinders entering obstacle_config as (r, r, h) boxes
        # (datasets/load_test_dataset.py:136-149); cylinder_config rows are [xyz, quat xyzw, radius, height] (:131-134)
        nb = self.n_obstacles - self.n_cylinders
        oc[nb:, 8] = oc[nb:, 7]
        cyl = np.concatenate([oc[nb:, :7], oc[nb:, 7:8], oc[nb:, 9:10]], axis=1) if self.n_cylinders else np.zeros((0, 9))
        return oc, oc[:nb].copy(), cyl, nb, self.n_cylinders, start, iks


# ---- neutral scene files -------------------------------------------------------------------------------------------
# The reference reads MPiNets problem pickles through geometrout objects (datasets/load_test_dataset.py:76-189): cuboids
# carry pose quaternions scalar-FIRST and are rolled to scalar-last (:126,:133); cylinders become boxes of extents
# (r, r, h) (:136-139, quirk Q9).  The pickles/geometrout are not available offline, so the same information is
# accepted as plain JSON: {"cuboids": [{"center": [x,y,z], "quaternion_wxyz": [w,x,y,z], "dims": [sx,sy,sz]}, ...],
# "cylinders": [{"center": ..., "quaternion_wxyz": ..., "radius": r, "height": h}, ...], "start": [7], "goals": [[7], ...]}.


def problem_to_arrays(problem: dict):
    """-> (obstacle_config (no,10), start (7,), goals (n,7)) in the fetch_data contract."""
    rows = []
    for c in problem.get("cuboids", []):
        w, x, y, z = c["quaternion_wxyz"]
        rows.append(np.concatenate([np.asarray(c["center"], float), [x, y, z, w], np.asarray(c["dims"], float)]))
    for c in problem.get("cylinders", []):
        w, x, y, z = c["quaternion_wxyz"]
        rows.append(cylinder_as_box(c["center"], [x, y, z, w], float(c["radius"]), float(c["height"])))
    if not rows:
        raise ValueError("scene has no obstacles")
    oc = np.stack(rows)
    start = np.asarray(problem["start"], dtype=np.float64)
    goals = np.atleast_2d(np.asarray(problem["goals"], dtype=np.float64))
    if oc.shape[1] != 10 or start.shape != (7,) or goals.shape[1] != 7:
        raise ValueError("malformed problem: need 10-column obstacles, 7-vector start, (n,7) goals")
    return oc, start, goals


def load_problem_file(path: str):
    import json

    with open(path) as f:
        return problem_to_arrays(json.load(f))


def save_problem_file(path: str, obstacle_config, start, goals) -> None:
    """inverse of load_problem_file for box obstacles."""
    import json

    cub = []
    for o in np.asarray(obstacle_config, float):
        x, y, z, w = o[3:7]
        cub.append({"center": o[:3].tolist(), "quaternion_wxyz": [w, x, y, z], "dims": o[7:10].tolist()})
    with open(path, "w") as f:
        json.dump({"cuboids": cub, "cylinders": [], "start": np.asarray(start, float).tolist(), "goals": np.atleast_2d(goals).tolist()}, f)


class ProblemSetDataset:
    """``TestDataset`` (datasets/load_test_dataset.py:15-189) over a problem-set JSON written by
    ``scripts/mpinets_pkl_to_json.py`` from an MPiNets pickle: same ``data_nums`` / ``fetch_data`` contract, same conversions
    (w-first -> w-last roll :126,:133; cuboids before cylinders :141-149; a cylinder row of ``obstacle_config`` is the box
    (r, r, h) :136-139; ``cylinder_config`` rows are [xyz, quat xyzw, radius, height] :131-134).  ``data_nums['merged_cubby']``
    is the length of the CUBBY list, like the reference's (:61).

    IK goals (robofin's ikfast on the problem's target pose, :170-187) are an explicit input: the problem's own ``goals`` list if
    the file carries one, else ``ik(target_xyz, target_quaternion_wxyz) -> (n, 7)`` if a callable is given; neither -> ValueError."""

    def __init__(self, path, ik=None):
        import json

        with open(path) as f:
            doc = json.load(f)
        if "scene_types" not in doc:
            raise ValueError(f"{path}: not a problem-set file (no 'scene_types'); a single problem loads with scenes.load_problem_file")
        self.problems = doc["scene_types"]
        self.ik = ik
        self.data_nums = {st: len(pr) for st, pr in self.problems.items()}
        if "merged_cubby" in self.data_nums and "cubby" in self.data_nums:
            self.data_nums["merged_cubby"] = len(self.problems["cubby"])  # datasets/load_test_dataset.py:61

    def fetch_data(self, scene_num, scene_type="tabletop"):
        if scene_type not in self.problems:
            raise ModuleNotFoundError(f"no scene type {scene_type!r} in this problem set ({sorted(self.problems)})")
        pr = self.problems[scene_type][scene_num]
        if "goals" in pr and len(pr["goals"]):
            goals = np.atleast_2d(np.asarray(pr["goals"], dtype=np.float64))
        elif self.ik is not None:
            goals = np.atleast_2d(np.asarray(self.ik(np.asarray(pr["target"]["xyz"], float), np.asarray(pr["target"]["quaternion_wxyz"], float)), dtype=np.float64))
        else:
            raise ValueError(f"{scene_type}[{scene_num}] carries no IK goals and no ik callable was given (robofin is not part of this package: "
                             "compute FrankaRobot.ik of the target elsewhere and pass --ik-goals to the converter, or ik= here)")
        oc, start, _ = problem_to_arrays({**pr, "goals": goals})
        nb, nc = len(pr.get("cuboids", [])), len(pr.get("cylinders", []))
        cub = oc[:nb].copy() if nb else []
        cyl = np.concatenate([oc[nb:, :7], oc[nb:, 7:8], oc[nb:, 9:10]], axis=1) if nc else []
        return oc, cub, cyl, nb, nc, start, goals
