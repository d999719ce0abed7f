"""CPU: host-side logic of the product package and the C-ABI surface (no compute calls without a GPU)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import yaml

from tests.util import T, cfgs_for

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_guide_cfgs_matches_reference_arrays(golden):
    from edmp_amd import guide_cfg as GC

    g = golden("g1_guide_cfgs")
    hyper = json.loads(str(g["hyper_json"]))
    # (a) from the reference's own parsed YAML content
    dicts = [{"hyperparameters": hyper[str(int(n))]} for n in g["guides"]]
    c = GC.build_guide_cfgs(dicts, int(g["bpg"]), T)
    # (b) from the built-in catalogue
    c2 = cfgs_for(g["guides"], g["bpg"])
    for k in ("clearance", "expansion", "guidance_method", "grad_norm", "guidance_schedule", "volume_trust_region"):
        assert np.array_equal(c[k], g[k]), k
        assert np.array_equal(c2[k], g[k]), k
    assert c["total_batch_size"] == len(g["guides"]) * int(g["bpg"])
    # guide18: isr3 [0,20) overwrites the tail of isr2 [10,40)
    i18 = list(g["guides"]).index(18) * int(g["bpg"])
    assert np.all(c["expansion"][i18, 10:20] == 0) and c["expansion"][i18, 20] > 0


def test_yaml_plugin_round_trip(tmp_path):
    from edmp_amd import guide_cfg as GC

    GC.write_guide_yamls(str(tmp_path))
    for n in GC.GUIDE_CATALOG:
        d = GC.load_guide_dict(n, str(tmp_path))
        assert d["hyperparameters"] == GC.catalog_guide_dict(n)["hyperparameters"]
    # a user-supplied guide file overrides the catalogue
    custom = GC.catalog_guide_dict(1)
    custom["hyperparameters"]["obstacle_clearance"]["range"] = [0.2, 0.3]
    with open(tmp_path / "cfgs" / "guide1.yaml", "w") as f:
        yaml.safe_dump(custom, f)
    run_cfg = {"guide": {"guides": [1, 10], "batch_size_per_guide": 3, "guide_path": str(tmp_path)}, "model": {"T": T}}
    c = GC.guide_cfgs_from_run_cfg(run_cfg)
    assert c["total_batch_size"] == 6 and c["clearance"][0, 0] == 0.2 and c["clearance"][0, -1] == 0.3
    assert c["guidance_method"].tolist() == [0, 0, 0, 1, 1, 1]
    with pytest.raises(FileNotFoundError):
        GC.load_guide_dict(6, str(tmp_path))  # guide6.yaml does not exist in the reference either


def test_split_rows_and_ragged_guides():
    from edmp_amd import guide_cfg as GC

    r = GC.split_rows(1024, 6)
    assert sum(r) == 1024 and max(r) - min(r) <= 1
    c = cfgs_for([1, 2, 3, 4, 5, 10], 0, rows_per_guide=r)
    assert c["total_batch_size"] == 1024 and int(c["guidance_method"].sum()) == r[-1]


def test_noise_stream_equals_reference_draw_order():
    """np.random.multivariate_normal(0, I_50, size=(B,7)) (diffusion.py:303,126) == standard_normal((B,7,50))."""
    from edmp_amd.diffusion import draw_noise

    B = 3
    np.random.seed(42)
    ref = [np.random.multivariate_normal(mean=np.zeros(50), cov=np.eye(50), size=(B, 7)) for _ in range(4)]
    np.random.seed(42)
    mine = draw_noise(3, B, 7, 50)
    assert mine.shape == (4, B, 7, 50)
    for i in range(4):
        assert np.array_equal(mine[i], ref[i])


def test_parallel_legacy_normal_stream_is_numpys_bit_for_bit():
    """edmp_amd.nprng (libedmp_nprng.so): the values AND the global RandomState afterwards are NumPy's own, for even / odd
    counts, with and without a cached second gaussian going in, across the multi-block threshold, with any thread count."""
    from edmp_amd import nprng

    if not nprng.available():
        pytest.skip("libedmp_nprng.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    block = 2 * (1 << 18)  # values one bulk block can yield at most
    for seed, warm, n in [(1, 0, 4096), (2, 3, 4097), (3, 4, 100_001), (4, 1, 2 * block + 12_345), (5, 0, 3 * block)]:
        np.random.seed(seed)
        np.random.standard_normal(warm)  # an odd warm-up leaves a cached gaussian in the state
        want = np.random.standard_normal(n)
        after = (np.random.standard_normal(5), np.random.random_sample(3), np.random.randint(0, 1 << 30, 4))
        for nthreads in (1, 3, None):
            np.random.seed(seed)
            np.random.standard_normal(warm)
            got = nprng.standard_normal(n, nthreads=nthreads)
            assert got.dtype == np.float64 and np.array_equal(got, want), (seed, n, nthreads)
            for a, b in zip(after, (np.random.standard_normal(5), np.random.random_sample(3), np.random.randint(0, 1 << 30, 4))):
                assert np.array_equal(a, b), (seed, n, nthreads, "state after the call")
    # shapes, and the small-size / foreign-bit-generator fallbacks stay NumPy's
    np.random.seed(7)
    a = np.random.standard_normal((3, 5, 7, 50))
    np.random.seed(7)
    assert np.array_equal(nprng.standard_normal((3, 5, 7, 50)), a)
    np.random.seed(8)
    b = np.random.standard_normal((40, 7, 50))
    np.random.seed(8)
    assert np.array_equal(nprng.standard_normal((40, 7, 50)), b)


def test_weights_inventory():
    from edmp_amd import weights as W

    s = W.unet_param_shapes()
    assert len(s) == 290 and sum(int(np.prod(v)) for v in s.values()) == 29_938_471
    sd = W.init_state_dict(3)
    assert W.infer_dims(sd) == (7, 32, (32, 64, 128, 256, 512, 512))
    assert sd["up_samplers.0.up.3.weight"].shape == (512, 512, 4)
    assert "down_samplers.5.down.3.weight" not in sd and "down_samplers.1.down.1.residual_conv.weight" not in sd
    sd2 = W.init_state_dict(3)
    assert all(np.array_equal(sd[k], sd2[k]) for k in sd)


def test_checkpoint_format_round_trip(tmp_path):
    from edmp_amd import weights as W

    dims = (16, 16, 32, 32, 64, 64)
    sd = W.init_state_dict(1, dims=dims)
    W.save_checkpoint_dir(str(tmp_path / "m"), sd)
    assert os.path.exists(tmp_path / "m" / "weights_latest.pt") and os.path.exists(tmp_path / "m" / "losses.npy")
    back = W.load_checkpoint_dir(str(tmp_path / "m"))
    assert list(back) == list(sd) and all(np.array_equal(back[k], sd[k]) for k in sd)


def test_checkpoint_written_by_the_reference():
    """G14: weights_latest.pt produced by the reference's own TemporalUNet.save() (temporalunet.py:78-86) for a tiny
    network (oracle/gen_golden_metrics.py): the reader returns every tensor of the state dict with the layout the
    packer expects, and the oracle UNet reproduces the reference forward stored next to it."""
    import torch

    from edmp_amd import weights as W
    from oracle import edmp_oracle as O

    d = os.path.join(os.path.dirname(__file__), "golden", "g14_ref_checkpoint")
    sd = W.load_checkpoint_dir(d)
    fw = np.load(os.path.join(d, "forward.npz"))
    dims = tuple(int(v) for v in fw["dims"])
    assert W.infer_dims(sd) == (7, 32, dims)
    shapes = W.unet_param_shapes(7, 32, dims)
    assert list(sd) == list(shapes) and all(tuple(sd[k].shape) == tuple(shapes[k]) and sd[k].dtype == np.float32 for k in shapes)
    y = O.UNetOracle(sd)(torch.tensor(fw["x"]), torch.tensor(fw["t"])).numpy()
    assert np.abs(y - fw["y"]).max() <= 1e-6


def test_packed_weight_file_format(tmp_path):
    """weights_packed.edmp: page-aligned float blob behind a header naming the architecture and the packing layout;
    truncated / foreign files are rejected (the loader then falls back to the state dict)."""
    from edmp_amd import weights as W

    blob = np.arange(1000, dtype=np.float32) * 0.5
    p = str(tmp_path / W.PACKED_NAME)
    W.write_packed(p, 201, 7, 32, (16, 16, 32), 50, 255, blob)
    r = W.read_packed(p)
    assert (r["layout"], r["input_dim"], r["time_dim"], r["dims"], r["horizon"], r["T"]) == (201, 7, 32, (16, 16, 32), 50, 255)
    assert np.array_equal(np.asarray(r["blob"]), blob) and r["blob"].ctypes.data % 4096 == 0
    with open(p, "r+b") as f:
        f.truncate(4096 + 100)
    assert W.read_packed(p) is None
    (tmp_path / "x.bin").write_bytes(b"not a packed file" * 400)
    assert W.read_packed(str(tmp_path / "x.bin")) is None and W.read_packed(str(tmp_path / "missing")) is None


def test_franka_tables():
    from edmp_amd import franka
    from oracle import edmp_oracle as O

    assert np.allclose(franka.static_frames(), np.array(O.STATIC_FRAMES, dtype=np.float32)[:, :3, :])
    dh = franka.dh_table()
    assert dh.shape == (7, 4) and dh[1, 2] != 0 and abs(dh[1, 2]) < 1e-7 and dh[1, 3] == -1.0  # f32 cos(pi/2) quirk
    he = franka.link_half_extents()
    assert np.allclose(he * 2, O.link_dimensions_effective(O.PLACEHOLDER_LINK_EXTENTS).numpy())
    lo, hi = franka.joint_limits()
    olo, ohi = O.joint_limits()
    assert np.array_equal(lo, olo) and np.array_equal(hi, ohi)
    with pytest.raises(ValueError):
        franka.link_half_extents(np.zeros((8, 3)))


def _write_g15_meshes(golden, d):
    g = golden("g15_link_meshes")
    for name, text in zip(g["link_names"], g["obj_texts"]):
        with open(os.path.join(str(d), str(name) + ".obj"), "w") as f:
            f.write(str(text))
    with open(os.path.join(str(d), "README.txt"), "w") as f:
        f.write("v 100 100 100\n")
    return g


def test_link_extents_read_from_meshes_like_the_reference(golden, tmp_path, monkeypatch):
    """a8 (lib/guide.py:245-282): G15 = nine non-box .obj files (40-200 vertices, vn / vt / f / comment lines, tabs, indented
    and 'v<TAB>' lines) and the link_dimensions / link_vertices the UNMODIFIED reference measured from them."""
    import warnings

    from edmp_amd import franka
    from oracle import edmp_oracle as O

    g = _write_g15_meshes(golden, tmp_path)
    ext = franka.link_extents_from_mesh_dir(tmp_path)
    assert ext.shape == (9, 3) and ext.dtype == np.float64
    dims = franka.link_half_extents(ext) * np.float32(2)
    assert np.array_equal(dims, g["link_dimensions"]), np.abs(dims - g["link_dimensions"]).max()
    assert np.array_equal(O.box_vertices(O.link_dimensions_effective(ext)).numpy(), g["link_vertices"])
    # lookup order: explicit table > mesh_dir > EDMP_MESH_DIR / pybullet_data > placeholder with ONE warning per process
    assert np.array_equal(franka.resolve_link_extents(np.ones((9, 3)), mesh_dir=tmp_path), np.ones((9, 3)))
    assert np.array_equal(franka.resolve_link_extents(None, mesh_dir=tmp_path), ext)
    monkeypatch.setenv("EDMP_MESH_DIR", str(tmp_path))
    assert np.array_equal(franka.resolve_link_extents(), ext)
    monkeypatch.delenv("EDMP_MESH_DIR")
    monkeypatch.setattr(franka, "default_mesh_dir", lambda: None)
    monkeypatch.setattr(franka, "_warned_placeholder", False)
    with pytest.warns(RuntimeWarning, match="PLACEHOLDER_LINK_EXTENTS"):
        assert np.array_equal(franka.resolve_link_extents(), franka.PLACEHOLDER_LINK_EXTENTS)
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # second call: silent
        franka.resolve_link_extents()
    # pybullet_data importable -> its directory is the default, like the reference
    import sys
    import types

    data = tmp_path / "pbdata" / "franka_panda" / "meshes" / "collision"
    data.mkdir(parents=True)
    _write_g15_meshes(golden, data)
    monkeypatch.undo()
    fake = types.ModuleType("pybullet_data")
    fake.getDataPath = lambda: str(tmp_path / "pbdata")
    monkeypatch.setitem(sys.modules, "pybullet_data", fake)
    monkeypatch.delenv("EDMP_MESH_DIR", raising=False)
    assert np.array_equal(franka.resolve_link_extents(), ext)
    # errors: a missing link file, a file without vertices
    os.remove(os.path.join(str(tmp_path), "hand.obj"))
    with pytest.raises(FileNotFoundError, match="hand.obj"):
        franka.link_extents_from_mesh_dir(tmp_path)
    with open(os.path.join(str(tmp_path), "hand.obj"), "w") as f:
        f.write("vn 1 2 3\nf 1 2 3\n")
    with pytest.raises(ValueError, match="no vertex"):
        franka.link_extents_from_mesh_dir(tmp_path)


def test_row_classes():
    from edmp_amd.guide import row_classes

    c = cfgs_for([1, 10, 1, 18], 3)
    rc, cc, ce = row_classes(c["clearance"], c["expansion"])
    assert rc.tolist() == [0, 0, 0, 1, 1, 1, 0, 0, 0, 2, 2, 2] and cc.shape == (3, T) and ce.shape == (3, T)


def test_row_classes_vectorised_form_equals_the_row_by_row_definition():
    """round 5: runs of equal rows are keyed once (the guide object is rebuilt per scene); any row order, NaN and signed-zero rows and
    a single row must give what the row-by-row dictionary gives: classes numbered by first appearance, representatives = first rows"""
    from edmp_amd.guide import row_classes

    def by_rows(clr, exp):
        keys, reps, rc = {}, [], []
        for b in range(clr.shape[0]):
            k = (clr[b].tobytes(), exp[b].tobytes())
            if k not in keys:
                keys[k] = len(reps)
                reps.append(b)
            rc.append(keys[k])
        return np.array(rc, dtype=np.int32), clr[reps], exp[reps]

    c = cfgs_for([1, 2, 3, 4, 5, 10], 171)
    clr, exp = np.asarray(c["clearance"], dtype=np.float64).copy(), np.asarray(c["expansion"], dtype=np.float64).copy()
    clr[7, 3], clr[8, 3] = np.nan, np.nan  # two equal NaN rows: one class (byte-wise keys)
    exp[500, 0], exp[501, 0] = 0.0, -0.0    # signed zeros differ byte-wise
    perm = np.random.RandomState(3).permutation(clr.shape[0])
    for a, b in ((clr, exp), (clr[perm], exp[perm]), (clr[:1], exp[:1]), (clr[5:9], exp[5:9])):
        got, want = row_classes(a, b), by_rows(a, b)
        assert np.array_equal(got[0], want[0]) and got[0].dtype == np.int32
        assert np.array_equal(got[1], want[1], equal_nan=True) and np.array_equal(got[2], want[2], equal_nan=True)


def test_total_rows_extension_of_the_run_config():
    """`guide.total_rows` (not in the reference's schema, which only allows B = G * batch_size_per_guide): BASELINE's 'batch = 1024 with
    six guides' as contiguous row blocks; absent -> the reference's arrays"""
    from edmp_amd import guide_cfg as GC

    run = {"guide": {"guides": [1, 2, 3, 4, 5, 10], "batch_size_per_guide": 170}, "model": {"T": T}}
    assert GC.guide_cfgs_from_run_cfg(run)["total_batch_size"] == 1020
    run["guide"]["total_rows"] = 1024
    c = GC.guide_cfgs_from_run_cfg(run)
    ref = GC.build_guide_cfgs([GC.catalog_guide_dict(n) for n in run["guide"]["guides"]], 170, T, rows_per_guide=GC.split_rows(1024, 6))
    assert c["total_batch_size"] == 1024 and all(np.array_equal(c[k], ref[k]) for k in ("clearance", "expansion", "guidance_method", "grad_norm", "guidance_schedule"))


def test_scene_contract():
    from edmp_amd import scenes

    s = scenes.random_scene(1, 8)
    assert s.shape == (8, 10) and np.allclose(np.linalg.norm(s[:, 3:7], axis=1), 1)
    assert np.array_equal(s, scenes.random_scene(1, 8))
    row = scenes.cylinder_as_box([0.5, 0, 0.2], [0, 0, 0, 1], 0.1, 0.4)
    assert row[7:].tolist() == [0.1, 0.1, 0.4]  # Q9: radius, not diameter


def test_c_abi_exports_every_declared_symbol():
    from edmp_amd import _capi

    hdr = open(os.path.join(ROOT, "include", "edmp_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(edmp_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    lib = _capi.load()  # dlopen works without a GPU
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.edmp_version() >= 100
    # struct layout the header promises
    assert ctypes.sizeof(_capi.UNetDesc) == 4 * (3 + 8 + 2)
    d = _capi.UNetDesc()
    d.input_dim, d.time_dim, d.n_levels, d.horizon, d.T = 7, 32, 6, 50, 255
    for i, v in enumerate((32, 64, 128, 256, 512, 512)):
        d.dims[i] = v
    assert lib.edmp_unet_param_count(ctypes.byref(d)) == 29_938_471  # host-only entry point


def test_product_fails_loudly_without_gpu():
    import torch

    from edmp_amd import _capi
    from edmp_amd.runtime import get_context

    with pytest.raises(_capi.EdmpError):
        get_context("cpu")
    if not torch.cuda.is_available():
        with pytest.raises(_capi.EdmpError):
            get_context("cuda:0")
        from edmp_amd.diffusion import Diffusion

        with pytest.raises(_capi.EdmpError):
            Diffusion(255, "cuda:0")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "edmp_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn
    drv = open(os.path.join(ROOT, "infer_serial.py")).read() if os.path.exists(os.path.join(ROOT, "infer_serial.py")) else ""
    assert not re.search(r"^\s*(from|import)\s+oracle", drv, flags=re.M)


def test_bench_refuses_a_multi_gpu_run_it_cannot_honour():
    """`bench.py --gpus N` must not be able to print a mislabelled single-rank line.  Without enough
    visible GPUs (none in the build container) and the RCCL backend it exits non-zero before anything is measured; a
    WORLD_SIZE that disagrees with --gpus is refused too.  (The re-exec under torch.distributed.run is a -m gpu test.)"""
    import subprocess
    import sys

    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the refusal path is not reachable here")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "EDMP_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr and "{" not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env={**env, "WORLD_SIZE": "4", "EDMP_DIST_BACKEND": "gloo"}, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr and "{" not in r.stdout


def test_infer_serial_job_summary_and_rank_detection(monkeypatch):
    """the scene-sharded driver's host logic: outside a launcher there is one rank on the cfg's device; the job summary adds the
    per-scene tallies up (the collective half runs in tests/test_dist_cpu.py)."""
    import infer_serial

    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert infer_serial._ranks("cuda:0") == (0, 1, "cuda:0")
    res = [dict(success_proxy=1, success_strict=0, rows_collision_free=3, rows=4, planning_time_s=0.5), dict(success_proxy=0, success_strict=0, rows_collision_free=0, rows=4, planning_time_s=0.25)]
    s = infer_serial.job_summary(res)
    assert s == dict(scenes=2, success_proxy=1, success_strict=0, rows_collision_free=3, rows=8, planning_time_s=0.75, ranks=1)


def test_pinned_noise_stream_watermark():
    """edmp_amd.diffusion.PinnedNoiseStream: the consumer blocks until the producer has published enough of the stream, and a producer error
    reaches the consumer instead of a hang (infer_serial's feeder thread -> Diffusion.denoise_guided's chunked upload)."""
    import threading
    import time

    import torch

    from edmp_amd.diffusion import PinnedNoiseStream

    st = PinnedNoiseStream(torch.zeros(8, dtype=torch.float64))
    seen = []

    def consumer():
        for n in (2, 5, 8):
            st.wait_until(n)
            seen.append((n, st.drawn))

    th = threading.Thread(target=consumer)
    th.start()
    for n in (1, 2, 6, 8):
        time.sleep(0.02)
        st.publish(n)
    th.join(timeout=5)
    assert not th.is_alive() and [s[0] for s in seen] == [2, 5, 8] and all(d >= n for n, d in seen)
    bad = PinnedNoiseStream(torch.zeros(4, dtype=torch.float64))
    bad.publish(1, error=RuntimeError("draw failed"))
    with pytest.raises(RuntimeError, match="draw failed"):
        bad.wait_until(3)
