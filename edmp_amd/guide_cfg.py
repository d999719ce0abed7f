"""Guide plugin surface: guide definitions -> per-row hyper-parameter arrays.

Mirrors the reference's config path: a run config (benchmark/cfgs/cfg*.yaml schema, cfg1.yaml:1-24) names guide
numbers; each guide is a YAML file ``<guide_path>/cfgs/guide<N>.yaml`` (schema guides/cfgs/guide1.yaml:1-20); the
driver expands them into per-row arrays ``guide_cfgs`` (infer_serial.py:56-91).  This module re-implements that
expansion (`build_guide_cfgs`), reads the same YAML schema with PyYAML (autolab_core is not needed), and carries the
catalogue of the reference's 16 shipped guides as data (`GUIDE_CATALOG`) so that runs do not
depend on a checkout of the reference; `write_guide_yamls` emits them in the reference's schema.
"""
from __future__ import annotations

import os

import numpy as np
import yaml

# (clearance range, [(isr, val) x3 in file order isr1, isr2, isr3], method, grad_norm, schedule type, scale_val)
_Z = (0.0, 0.0)


def _g(clr, e1, e2, e3, method, gn, stype, scale):
    return dict(clearance=clr, expansion=(e1, e2, e3), method=method, grad_norm=gn, sched=stype, scale=scale)


_E_NONE = (((150, 255), _Z), ((20, 150), _Z), ((0, 20), _Z))
_E_UP = (((150, 255), (0.4, 0.4)), ((20, 150), (0.0, 0.4)), ((0, 20), _Z))
_E_DOWN = (((150, 255), (0.4, 0.4)), ((20, 150), (0.4, 0.0)), ((0, 20), _Z))
_E_10 = (((80, 255), (0.4, 0.4)), ((20, 80), _Z), ((0, 20), _Z))
_E_18 = (((40, 255), (0.4, 0.4)), ((10, 40), (0.0, 0.4)), ((0, 20), _Z))

GUIDE_CATALOG = {
    1: _g((0.1, 0.1), *_E_NONE, "iv", False, "varying", 0.05),
    2: _g((0.05, 0.05), *_E_NONE, "iv", False, "varying", 0.05),
    3: _g((0.01, 0.01), *_E_NONE, "iv", False, "varying", 0.05),
    4: _g((0.15, 0.15), *_E_NONE, "iv", False, "varying", 0.05),
    5: _g((0.01, 0.15), *_E_NONE, "iv", False, "varying", 0.05),
    9: _g(_Z, *_E_DOWN, "iv", True, "constant", 0.05),
    10: _g((0.06, 0.06), *_E_10, "sv", False, "varying", 0.05),
    11: _g(_Z, *_E_UP, "sv", True, "constant", 0.05),
    12: _g(_Z, *_E_DOWN, "iv", True, "constant", 0.05),
    13: _g(_Z, *_E_UP, "sv", True, "constant", 0.01),
    14: _g((0.02, 0.02), *_E_UP, "sv", True, "constant", 0.1),
    15: _g(_Z, *_E_DOWN, "iv", True, "constant", 0.05),
    16: _g((0.1, 0.1), *_E_UP, "sv", True, "constant", 0.1),
    17: _g(_Z, *_E_DOWN, "iv", True, "constant", 0.05),
    18: _g((0.05, 0.05), *_E_18, "sv", True, "constant", 0.05),
    21: _g((0.05, 0.05), *_E_18, "sv", True, "constant", 0.1),
}

VOLUME_TRUST_REGION = 0.0008  # every shipped guide; the driver hard-codes the same value (infer_serial.py:125)


def catalog_guide_dict(n: int) -> dict:
    """The dict a reference-schema guide<N>.yaml parses to."""
    g = GUIDE_CATALOG[n]
    oe = {}
    for k, (isr, val) in zip("123", g["expansion"]):
        oe["isr" + k] = list(isr)
        oe["val" + k] = [float(val[0]), float(val[1])]
    return {
        "index": n,
        "hyperparameters": {
            "obstacle_clearance": {"range": [float(g["clearance"][0]), float(g["clearance"][1])]},
            "obstacle_expansion": oe,
            "guidance_method": g["method"],
            "grad_norm": bool(g["grad_norm"]),
            "guidance_schedule": {"type": g["sched"], "scale_val": float(g["scale"])},
            "volume_trust_region": VOLUME_TRUST_REGION,
        },
    }


def write_guide_yamls(guide_path: str, guides=None) -> None:
    """Emit ``<guide_path>/cfgs/guide<N>.yaml`` in the reference's schema."""
    d = os.path.join(guide_path, "cfgs")
    os.makedirs(d, exist_ok=True)
    for n in guides or sorted(GUIDE_CATALOG):
        with open(os.path.join(d, f"guide{n}.yaml"), "w") as f:
            yaml.safe_dump(catalog_guide_dict(n), f, sort_keys=False)


def load_yaml(path: str) -> dict:
    with open(path) as f:
        return dict(yaml.safe_load(f))


def load_guide_dict(n: int, guide_path: str | None = None) -> dict:
    """guide<N>.yaml from ``guide_path`` if it exists there (user plugin), else the built-in catalogue."""
    if guide_path is not None:
        p = os.path.join(guide_path, "cfgs", f"guide{n}.yaml")
        if os.path.exists(p):
            return load_yaml(p)
    if n not in GUIDE_CATALOG:
        raise FileNotFoundError(f"guide{n}.yaml not found under {guide_path!r} and not in the built-in catalogue")
    return catalog_guide_dict(n)


def build_guide_cfgs(guide_dicts, batch_size_per_guide, T: int, rows_per_guide=None) -> dict:
    """Per-row arrays exactly as infer_serial.py:56-91 builds them.

    ``rows_per_guide`` (optional list) lets guide i own an arbitrary number of contiguous rows (needed for
    "B = 1024 with 6 guides"); default is the reference's ``batch_size_per_guide`` for every guide.
    """
    G = len(guide_dicts)
    counts = [int(batch_size_per_guide)] * G if rows_per_guide is None else [int(c) for c in rows_per_guide]
    B = int(sum(counts))
    cfgs = {
        "batch_size_per_guide": batch_size_per_guide,
        "total_batch_size": B,
        "clearance": np.zeros((B, T)),
        "expansion": np.zeros((B, T)),
        "guidance_method": np.zeros((B,)),
        "grad_norm": np.zeros((B,)),
        "guidance_schedule": np.zeros((B, T)),
        "volume_trust_region": np.zeros((B,)),
    }
    r0 = 0
    for g, cnt in zip(guide_dicts, counts):
        rows = slice(r0, r0 + cnt)
        r0 += cnt
        h = g["hyperparameters"]
        rng = h["obstacle_clearance"]["range"]
        cfgs["clearance"][rows, :] = np.linspace(rng[0], rng[1], T)
        oe = h["obstacle_expansion"]
        for k in "123":  # isr1, isr2, isr3 in this order: later segments overwrite earlier ones
            lo, hi = oe["isr" + k]
            v0, v1 = oe["val" + k]
            cfgs["expansion"][rows, lo:hi] = np.linspace(v0, v1, num=abs(hi - lo))
        cfgs["guidance_method"][rows] = 1 if h["guidance_method"] == "sv" else 0
        cfgs["grad_norm"][rows] = 1 if h["grad_norm"] else 0
        gs = h["guidance_schedule"]
        cfgs["guidance_schedule"][rows, :] = (1.4 + np.arange(T) / T) if gs["type"] == "varying" else gs["scale_val"]
        cfgs["volume_trust_region"][rows] = h["volume_trust_region"]
    return cfgs


def split_rows(total: int, n_guides: int):
    """guide g owns rows [floor(g*total/G), floor((g+1)*total/G))."""
    edges = [(g * total) // n_guides for g in range(n_guides + 1)]
    return [edges[g + 1] - edges[g] for g in range(n_guides)]


def guide_cfgs_from_run_cfg(run_cfg: dict, base_dir: str = ".") -> dict:
    """``run_cfg`` = parsed benchmark cfg (cfg1.yaml schema).  Resolves guides and builds the row arrays."""
    gpath = run_cfg["guide"].get("guide_path")
    if gpath is not None and not os.path.isabs(gpath):
        gpath = os.path.join(base_dir, gpath)
    dicts = [load_guide_dict(int(n), gpath) for n in run_cfg["guide"]["guides"]]
    # extension (not in the reference's schema, which only allows B = G * batch_size_per_guide): `total_rows: 1024` deals that many
    # rows to the guides as contiguous blocks (split_rows) - BASELINE.json's "batch=1024, 6-guide ensemble"
    total = run_cfg["guide"].get("total_rows")
    rows = split_rows(int(total), len(dicts)) if total else None
    return build_guide_cfgs(dicts, run_cfg["guide"]["batch_size_per_guide"], int(run_cfg["model"]["T"]), rows_per_guide=rows)
