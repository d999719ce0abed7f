#!/usr/bin/env python
"""mpinets_pkl_to_json.py — turn an MPiNets problem pickle (`global_solvable_problems.pkl`, `hybrid_solvable_problems.pkl`,
`both_solvable_problems.pkl`: what the reference's TestDataset opens, datasets/load_test_dataset.py:15-63) into the neutral
JSON problem-set file `edmp_amd.scenes.ProblemSetDataset` / `scenes.load_problem_file` read.

    python scripts/mpinets_pkl_to_json.py datasets/global_solvable_problems.pkl global_problems.json [--ik-goals goals.json]

The pickles hold `mpinets.types.PlanningProblem` dataclasses (mpinets/types.py:35-46) whose obstacles are
`geometrout.primitive.Cuboid` / `Cylinder` objects.  Neither geometrout nor robofin nor mpinets needs to be installed: a
restricted `pickle.Unpickler` maps every class of those packages (and pyquaternion, which geometrout's SO3 wraps) to a light
attribute record, lets an exact list of NumPy's array / scalar / dtype reconstruction helpers through (`_NUMPY_GLOBALS`; NOT
`numpy.*`, which also holds `exec`-like helpers) and refuses every other global a pickle names.  The converter then applies the reference loader's conversions (datasets/load_test_dataset.py):
  * the three problem types of a scene type are concatenated task_oriented | neutral_start | neutral_goal (:53-56);
  * quaternions are stored scalar-FIRST in the pickles (`list(obstacle._pose._so3._quat)`, :108, :114) - kept as
    `quaternion_wxyz` in the JSON; the w-first -> w-last roll of :126 / :133 happens in `scenes.problem_to_arrays`;
  * cuboids come before cylinders in obstacle_config (:141-149), a cylinder enters the guide as a box of extents (r, r, h)
    (:136-139, quirk Q9) - `problem_to_arrays` again, from the `radius` / `height` kept here;
  * obstacles that are neither Cuboid nor Cylinder are ignored, as the reference's isinstance chain ignores them (:105-116).
IK goals (robofin's ikfast on `data.target`, :170-187) cannot be computed here: the end-effector target pose is written out
(`target`), and goals are an explicit input - `--ik-goals file.json` = {"<scene_type>": [[[7 floats], ...] per problem]} - or
are supplied at load time (`ProblemSetDataset(path, ik=callable)`).
"""
from __future__ import annotations

import argparse
import io
import json
import pickle
import sys

import numpy as np

STUB_PACKAGES = ("geometrout", "mpinets", "pyquaternion", "robofin")
PROBLEM_TYPES = ("task_oriented", "neutral_start", "neutral_goal")  # datasets/load_test_dataset.py:53-56
_SAFE_BUILTINS = {"list", "dict", "set", "frozenset", "tuple", "complex", "bytearray", "slice", "range", "object", "float", "int", "str", "bool", "bytes"}


class Record:
    """attribute bag standing for an instance of a class we do not import; remembers the class it stood for."""

    _cls_module = "?"
    _cls_name = "?"

    def __init__(self, *args, **kwargs):
        self._init_args = args
        self.__dict__.update(kwargs)

    def __setstate__(self, state):
        # object.__reduce_ex__ states: a dict, or (dict, slots-dict)
        if isinstance(state, tuple) and len(state) == 2:
            for part in state:
                if isinstance(part, dict):
                    self.__dict__.update(part)
        elif isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__["_state"] = state

    def __repr__(self):
        return f"<{self._cls_module}.{self._cls_name} {sorted(k for k in self.__dict__ if not k.startswith('_init'))}>"


_stub_cache: dict = {}


def _stub(module: str, name: str):
    key = (module, name)
    if key not in _stub_cache:
        _stub_cache[key] = type(name, (Record,), {"_cls_module": module, "_cls_name": name})
    return _stub_cache[key]


# the NumPy globals an ndarray / scalar / dtype pickle names (protocols 2-5, NumPy 1.x `numpy.core` and 2.x `numpy._core` spellings);
# everything else under numpy.* is refused
_NUMPY_GLOBALS = frozenset(
    (mod, name)
    for mod in ("numpy", "numpy.core.multiarray", "numpy._core.multiarray", "numpy.core.numeric", "numpy._core.numeric")
    for name in ("_reconstruct", "scalar", "ndarray", "dtype", "_frombuffer")
)


class StubUnpickler(pickle.Unpickler):
    """geometrout / mpinets / pyquaternion / robofin classes -> Record stubs; NumPy's array / scalar / dtype reconstruction helpers (an exact
    (module, name) list) and plain builtins -> the real things; anything else is refused."""

    def find_class(self, module, name):
        top = module.split(".")[0]
        if top in STUB_PACKAGES:
            return _stub(module, name)
        if (module, name) in _NUMPY_GLOBALS:  # exact pairs only: `numpy.*` as a whole holds callables that run code (numpy.testing..runstring)
            return super().find_class(module, name)
        if module == "builtins" and name in _SAFE_BUILTINS:
            return super().find_class(module, name)
        if (module, name) in (("collections", "OrderedDict"), ("collections", "defaultdict"), ("copyreg", "_reconstructor"), ("copy_reg", "_reconstructor"),
                              ("dataclasses", "_HAS_DEFAULT_FACTORY_CLASS"),
                              ("_codecs", "encode")):  # (protocol-2 pickles carry an array's bytes as a latin-1 string)
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refusing to unpickle {module}.{name}: not a geometrout / mpinets / numpy type")


def load_pickle(path_or_bytes):
    if isinstance(path_or_bytes, (bytes, bytearray)):
        return StubUnpickler(io.BytesIO(path_or_bytes)).load()
    with open(path_or_bytes, "rb") as f:
        return StubUnpickler(f).load()


# ---- attribute access that survives the small layout differences between geometrout releases -----------------------------
def _get(obj, *names):
    for n in names:
        if isinstance(obj, dict) and n in obj:
            return obj[n]
        if hasattr(obj, "__dict__") and n in obj.__dict__:
            return obj.__dict__[n]
    raise KeyError(f"none of {names} in {obj!r}")


def _vec(x, n):
    a = np.asarray(getattr(x, "q", x) if not isinstance(x, (list, tuple, np.ndarray)) else x, dtype=np.float64).reshape(-1)
    if a.shape[0] != n:
        raise ValueError(f"expected {n} numbers, got {a.shape[0]}: {x!r}")
    return [float(v) for v in a]


def _pose_parts(pose):
    """SE3 record -> (xyz, quaternion w-first).  geometrout 0.0.3.x: SE3(_xyz, _so3), SO3(_quat = pyquaternion.Quaternion(q))."""
    xyz = _vec(_get(pose, "_xyz", "xyz", "pos", "_pos"), 3)
    so3 = _get(pose, "_so3", "so3")
    quat = _get(so3, "_quat", "quat", "q", "_q") if isinstance(so3, Record) else so3
    if isinstance(quat, Record):  # pyquaternion.Quaternion: state {'q': array([w, x, y, z])}
        quat = _get(quat, "q", "_q")
    return xyz, _vec(quat, 4)


def obstacle_to_dict(ob):
    """-> ('cuboid' | 'cylinder' | None, dict).  center = the pose's translation (geometrout's `center` property)."""
    kind = getattr(ob, "_cls_name", type(ob).__name__)
    if kind not in ("Cuboid", "Cylinder"):
        return None, None
    xyz, quat = _pose_parts(_get(ob, "_pose", "pose"))
    if "center" in ob.__dict__ or "_center" in ob.__dict__:
        xyz = _vec(_get(ob, "center", "_center"), 3)
    if kind == "Cuboid":
        return "cuboid", {"center": xyz, "quaternion_wxyz": quat, "dims": _vec(_get(ob, "dims", "_dims"), 3)}
    return "cylinder", {"center": xyz, "quaternion_wxyz": quat, "radius": float(np.asarray(_get(ob, "radius", "_radius")).reshape(())),
                        "height": float(np.asarray(_get(ob, "height", "_height")).reshape(()))}


def problem_to_dict(pr, goals=None):
    cub, cyl = [], []
    for ob in (_get(pr, "obstacles") or []):
        kind, d = obstacle_to_dict(ob)
        if kind == "cuboid":
            cub.append(d)
        elif kind == "cylinder":
            cyl.append(d)
    txyz, tquat = _pose_parts(_get(pr, "target"))
    out = {"cuboids": cub, "cylinders": cyl, "start": _vec(_get(pr, "q0"), 7), "target": {"xyz": txyz, "quaternion_wxyz": tquat, "frame": "right_gripper"}}
    if goals is not None:
        g = np.atleast_2d(np.asarray(goals, dtype=np.float64))
        if g.shape[1] != 7:
            raise ValueError("IK goals must be (n, 7)")
        out["goals"] = g.tolist()
    return out


def convert(data, ik_goals=None) -> dict:
    """ProblemSet (dict scene_type -> dict problem_type -> list of PlanningProblem records) -> the JSON document."""
    doc = {"format": "edmp_amd problem set v1", "source": "MPiNets ProblemSet pickle via scripts/mpinets_pkl_to_json.py",
           "order": "task_oriented | neutral_start | neutral_goal (datasets/load_test_dataset.py:53-56)", "scene_types": {}}
    for scene_type, by_type in data.items():
        problems = []
        for pt in PROBLEM_TYPES:
            problems.extend(list(by_type.get(pt, [])))
        goals = (ik_goals or {}).get(scene_type)
        if goals is not None and len(goals) != len(problems):
            raise ValueError(f"{scene_type}: {len(goals)} IK-goal sets for {len(problems)} problems")
        doc["scene_types"][scene_type] = [problem_to_dict(p, None if goals is None else goals[i]) for i, p in enumerate(problems)]
    return doc


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("pickle")
    ap.add_argument("json")
    ap.add_argument("--ik-goals", help='JSON {"<scene_type>": [ [[7 floats], ...] per problem ]} (robofin IK of each target, computed elsewhere)')
    args = ap.parse_args(argv)
    goals = json.load(open(args.ik_goals)) if args.ik_goals else None
    doc = convert(load_pickle(args.pickle), goals)
    with open(args.json, "w") as f:
        json.dump(doc, f)
    for st, pr in doc["scene_types"].items():
        nb = sum(len(p["cuboids"]) for p in pr)
        nc = sum(len(p["cylinders"]) for p in pr)
        print(f"{st}: {len(pr)} problems, {nb} cuboids, {nc} cylinders, IK goals {'present' if pr and 'goals' in pr[0] else 'NOT included (supply at load time)'}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
