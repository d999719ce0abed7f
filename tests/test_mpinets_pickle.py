"""CPU: scripts/mpinets_pkl_to_json.py + scenes.ProblemSetDataset - the reference's real problem files
(MPiNets `*_solvable_problems.pkl`, datasets/load_test_dataset.py:15-63) become loadable without geometrout / robofin / mpinets.

The pickles are produced HERE from throw-away classes registered under the real module paths with the attribute layout the
reference's loader reads (`obstacle.center`, `.dims`, `.radius`, `.height`, `._pose._so3._quat`; `data.obstacles`, `.q0`,
`.target`; mpinets/types.py:35-46), then those modules are removed again: the converter must work with none of them importable.
Expected arrays are restated from the reference loader's arithmetic (roll :126,:133; (r, r, h) :136-139; order :141-149)."""
import importlib
import json
import os
import pickle
import sys
import types
from dataclasses import dataclass, field

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_converter():
    spec = importlib.util.spec_from_file_location("mpinets_pkl_to_json", os.path.join(ROOT, "scripts", "mpinets_pkl_to_json.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _fake_modules():
    """geometrout.primitive / geometrout.transform / mpinets.types / pyquaternion look-alikes (same names, same attributes)."""
    mods = {n: types.ModuleType(n) for n in ("geometrout", "geometrout.primitive", "geometrout.transform", "mpinets", "mpinets.types", "pyquaternion",
                                             "pyquaternion.quaternion")}

    class Quaternion:
        def __init__(self, q):
            self.q = np.asarray(q, dtype=np.float64)

        def __iter__(self):
            return iter(self.q)

    class SO3:
        def __init__(self, quat):
            self._quat = Quaternion(quat)

    class SE3:
        def __init__(self, xyz, quat):
            self._xyz = np.asarray(xyz, dtype=np.float64)
            self._so3 = SO3(quat)

        @property
        def xyz(self):
            return self._xyz

    class Cuboid:
        def __init__(self, center, dims, quaternion):
            self._pose = SE3(center, quaternion)
            self._dims = np.asarray(dims, dtype=np.float64)

        @property
        def center(self):
            return self._pose.xyz

        @property
        def dims(self):
            return self._dims

    class Cylinder:
        def __init__(self, center, radius, height, quaternion):
            self._pose = SE3(center, quaternion)
            self.radius = radius
            self.height = height

        @property
        def center(self):
            return self._pose.xyz

    class Sphere:
        def __init__(self, center, radius):
            self.center = np.asarray(center, float)
            self.radius = radius

    @dataclass
    class PlanningProblem:
        target: object
        target_volume: object
        q0: np.ndarray
        obstacles: list = None
        obstacle_point_cloud: np.ndarray = None
        target_negative_volumes: list = field(default_factory=lambda: [])

    for cls, mod in ((Quaternion, "pyquaternion.quaternion"), (SO3, "geometrout.transform"), (SE3, "geometrout.transform"), (Cuboid, "geometrout.primitive"),
                     (Cylinder, "geometrout.primitive"), (Sphere, "geometrout.primitive"), (PlanningProblem, "mpinets.types")):
        cls.__module__ = mod
        cls.__qualname__ = cls.__name__
        setattr(mods[mod], cls.__name__, cls)
    return mods


def _problem(ns, rs, n_cub, n_cyl):
    P, T = ns["geometrout.primitive"], ns["geometrout.transform"]
    unit = lambda q: q / np.linalg.norm(q)  # noqa: E731
    obs = []
    for _ in range(n_cyl):  # cylinders FIRST in the pickle: the loader must still put cuboids first
        obs.append(P.Cylinder(rs.uniform(-1, 1, 3), float(rs.uniform(0.02, 0.2)), float(rs.uniform(0.1, 0.5)), unit(rs.standard_normal(4))))
    obs.append(P.Sphere(rs.uniform(-1, 1, 3), 0.1))  # ignored by the reference's isinstance chain
    for _ in range(n_cub):
        obs.append(P.Cuboid(rs.uniform(-1, 1, 3), rs.uniform(0.05, 0.4, 3), unit(rs.standard_normal(4))))
    tgt = T.SE3(rs.uniform(0.2, 0.6, 3), unit(rs.standard_normal(4)))
    return ns["mpinets.types"].PlanningProblem(target=tgt, target_volume=obs[-1], q0=rs.uniform(-1, 1, 7), obstacles=obs)


def _expected(pr):
    """the reference loader's arrays for one PlanningProblem (datasets/load_test_dataset.py:92-149), restated"""
    cub = [o for o in pr.obstacles if type(o).__name__ == "Cuboid"]
    cyl = [o for o in pr.obstacles if type(o).__name__ == "Cylinder"]
    rows = [np.concatenate([o.center, np.roll(np.array(list(o._pose._so3._quat)), -1), o.dims]) for o in cub]
    rows += [np.concatenate([o.center, np.roll(np.array(list(o._pose._so3._quat)), -1), [o.radius, o.radius, o.height]]) for o in cyl]
    cylc = [np.concatenate([o.center, np.roll(np.array(list(o._pose._so3._quat)), -1), [o.radius, o.height]]) for o in cyl]
    return np.array(rows), (np.array(cylc) if cylc else []), len(cub), len(cyl)


@pytest.fixture()
def problem_pickle(tmp_path):
    mods = _fake_modules()
    saved = {n: sys.modules.get(n) for n in mods}
    sys.modules.update(mods)
    try:
        rs = np.random.RandomState(0)
        data = {st: {pt: [_problem(mods, rs, 2 + k, k % 3) for k in range(n)] for pt, n in (("task_oriented", 2), ("neutral_start", 1), ("neutral_goal", 3))}
                for st in ("tabletop", "cubby", "merged_cubby", "dresser")}
        data["merged_cubby"]["neutral_goal"].append(_problem(mods, rs, 3, 1))  # one more than cubby: the data_nums quirk shows
        path = tmp_path / "global_solvable_problems.pkl"
        with open(path, "wb") as f:
            pickle.dump(data, f)
        flat = {st: [p for pt in ("task_oriented", "neutral_start", "neutral_goal") for p in data[st][pt]] for st in data}
        expected = {st: [(_expected(p), np.asarray(p.q0, float), np.asarray(p.target._xyz, float), np.array(list(p.target._so3._quat))) for p in flat[st]] for st in flat}
    finally:
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m
    for n in mods:
        assert n not in sys.modules or saved[n] is not None
    return str(path), expected


def test_pickle_to_json_to_fetch_data_contract(problem_pickle, tmp_path):
    from edmp_amd import scenes

    conv = _load_converter()
    pkl, expected = problem_pickle
    with pytest.raises(Exception):  # the plain unpickler cannot even import the classes here
        pickle.load(open(pkl, "rb"))
    out = tmp_path / "problems.json"
    assert conv.main([pkl, str(out)]) == 0
    doc = json.load(open(out))
    assert sorted(doc["scene_types"]) == ["cubby", "dresser", "merged_cubby", "tabletop"] and len(doc["scene_types"]["tabletop"]) == 6
    ik_calls = []

    def ik(xyz, quat_wxyz):  # IK goals are an explicit input: here a stand-in that records the target it was asked for
        ik_calls.append((xyz.copy(), quat_wxyz.copy()))
        return np.tile(np.linspace(-0.5, 0.5, 7), (3, 1)) + len(ik_calls) * 1e-3

    ds = scenes.ProblemSetDataset(str(out), ik=ik)
    assert ds.data_nums == {"tabletop": 6, "cubby": 6, "merged_cubby": 6, "dresser": 6}  # merged_cubby reports cubby's length (load_test_dataset.py:61)
    for st in ("tabletop", "merged_cubby"):
        for i, ((oc_ref, cyl_ref, nb, nc), q0, txyz, tquat) in enumerate(expected[st][:6]):
            oc, cub, cyl, nb2, nc2, start, goals = ds.fetch_data(i, st)
            assert (nb2, nc2) == (nb, nc) and oc.shape == (nb + nc, 10)
            assert np.array_equal(oc, oc_ref) and np.array_equal(start, q0)
            assert np.array_equal(np.asarray(cub), oc_ref[:nb])
            assert (nc == 0 and len(cyl) == 0) or np.array_equal(cyl, cyl_ref)
            assert goals.shape == (3, 7) and np.array_equal(ik_calls[-1][0], txyz) and np.array_equal(ik_calls[-1][1], tquat)
    with pytest.raises(ValueError):
        scenes.ProblemSetDataset(str(out)).fetch_data(0, "tabletop")  # no goals in the file, no ik callable
    with pytest.raises(ModuleNotFoundError):
        ds.fetch_data(0, "kitchen")
    # goals baked into the file by the converter
    goals = {st: [np.full((2, 7), 0.1 * k).tolist() for k in range(len(doc["scene_types"][st]))] for st in doc["scene_types"]}
    gj = tmp_path / "goals.json"
    json.dump(goals, open(gj, "w"))
    out2 = tmp_path / "problems_with_goals.json"
    assert conv.main([pkl, str(out2), "--ik-goals", str(gj)]) == 0
    g = scenes.ProblemSetDataset(str(out2)).fetch_data(4, "dresser")[6]
    assert g.shape == (2, 7) and np.allclose(g, 0.4)
    # one problem of the set is also a valid single-problem file for scenes.load_problem_file
    single = tmp_path / "one.json"
    json.dump(json.load(open(out2))["scene_types"]["tabletop"][1], open(single, "w"))
    oc1, s1, g1 = scenes.load_problem_file(str(single))
    assert np.array_equal(oc1, expected["tabletop"][1][0][0]) and g1.shape == (2, 7)


def test_unpickler_refuses_foreign_code(tmp_path):
    conv = _load_converter()

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned",))

    with pytest.raises(pickle.UnpicklingError):
        conv.load_pickle(pickle.dumps({"tabletop": {"task_oriented": [Evil()]}}))


def test_unpickler_refuses_numpy_code_gadgets(tmp_path):
    """`numpy.*` as a whole is not safe to resolve - numpy.testing._private.utils.runstring(code, dict) is `exec`.  Only
    the exact reconstruction helpers of arrays / scalars / dtypes pass; real arrays of every pickle protocol still load."""
    conv = _load_converter()
    marker = tmp_path / "pwned"

    class Gadget:
        def __reduce__(self):
            from numpy.testing._private.utils import runstring

            return (runstring, (f"open({str(marker)!r}, 'w').write('x')", {}))

    with pytest.raises(pickle.UnpicklingError):
        conv.load_pickle(pickle.dumps({"tabletop": {"task_oriented": [Gadget()]}}))
    assert not marker.exists()
    for mod, name in (("numpy", "load"), ("numpy.lib.npyio", "load"), ("numpy.ctypeslib", "load_library"), ("numpy.f2py", "compile")):
        blob = b"\x80\x02c" + mod.encode() + b"\n" + name.encode() + b"\n."  # protocol-2 GLOBAL opcode naming (mod, name)
        with pytest.raises(pickle.UnpicklingError):
            conv.load_pickle(blob)
    payload = {"a": np.arange(12.0).reshape(3, 4), "s": np.float64(2.5), "i": np.int32(7), "f": np.asfortranarray(np.eye(3, dtype=np.float32))}
    for proto in range(2, pickle.HIGHEST_PROTOCOL + 1):
        back = conv.load_pickle(pickle.dumps(payload, protocol=proto))
        assert np.array_equal(back["a"], payload["a"]) and back["s"] == 2.5 and back["i"] == 7 and np.array_equal(back["f"], payload["f"])


def test_infer_serial_names_the_missing_converted_file(tmp_path):
    """a run config with the reference's dataset types ('global' | 'hybrid' | 'both', datasets/load_test_dataset.py:15-38) looks for the
    converted JSON next to where the pickle would be and says what to do when it is not there - before any GPU is touched"""
    import yaml

    import infer_serial

    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "cfg_c1_plumbing.yaml")))
    cfg["dataset"]["dataset_type"], cfg["dataset"]["path"] = "hybrid", str(tmp_path)
    os.makedirs(tmp_path / "cfgs")
    yaml.safe_dump(cfg, open(tmp_path / "cfgs" / "c.yaml", "w"))
    with pytest.raises(FileNotFoundError, match="mpinets_pkl_to_json"):
        infer_serial.run(str(tmp_path / "cfgs" / "c.yaml"), verbose=False)
