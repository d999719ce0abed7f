"""G13 + G14 fixtures from the UNMODIFIED reference — TEST INFRASTRUCTURE ONLY (build container; needs /root/reference).

    python -m oracle.gen_golden_metrics

G13  lib/metrics.py:11-45 `MetricsCalculator.smoothness_metric` / `path_length_metric` (end-effector positions through
     the reference's own 10-row DH chain, lib/guide.py:100-116) and mpinets/third_party/sparc.py on six (7, 50)
     trajectories -> tests/golden/g13_metrics.npz (inputs + the reference's outputs; edmp_amd/evaluation.py is held to it).
G14  a checkpoint WRITTEN BY THE REFERENCE ITSELF: TemporalUNet.save() (diffusion/models/temporalunet.py:78-86) of a
     tiny network -> tests/golden/g14_ref_checkpoint/weights_latest.pt (+ the input / output of the reference's forward
     with those weights), the file edmp_amd.weights.load_checkpoint_dir must read.
Fixtures are data only (arrays, a torch state-dict file); no reference source text is stored.
"""
from __future__ import annotations

import os
import shutil
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from edmp_amd import guide_cfg as GC  # noqa: E402
from edmp_amd import scenes as SC  # noqa: E402
from oracle import edmp_oracle as O  # noqa: E402
from oracle import ref_harness  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
TINY_DIMS = (16, 16, 32, 32, 64, 64)


def trajectories():
    """six (7, 50) joint trajectories: straight line, smooth cubic, jerky, noisy line, constant, large-angle sweep"""
    rs = np.random.RandomState(13)
    s, g = SC.DEFAULT_START, SC.DEFAULT_GOAL
    u = np.linspace(0.0, 1.0, 50)
    line = s[:, None] + (g - s)[:, None] * u[None, :]
    cubic = s[:, None] + (g - s)[:, None] * (3 * u**2 - 2 * u**3)[None, :]
    jerky = line + 0.3 * np.sign(np.sin(40 * u))[None, :] * rs.rand(7, 1)
    noisy = line + 0.05 * rs.standard_normal((7, 50))
    const = np.repeat(s[:, None], 50, axis=1)
    sweep = np.stack([np.linspace(-2.5, 2.5, 50) * (1 + 0.1 * j) * (-1) ** j for j in range(7)])
    return np.stack([line, cubic, jerky, noisy, const, sweep]).astype(np.float64)


def main():
    torch.manual_seed(0)
    refd, refg = ref_harness.install(O.PLACEHOLDER_LINK_EXTENTS)
    import lib.metrics as ref_metrics  # noqa: E402  (the unmodified reference)
    from mpinets.third_party.sparc import sparc as ref_sparc  # noqa: E402

    cfgs = GC.build_guide_cfgs([GC.catalog_guide_dict(1)], 2, 255)
    guide = refg.IntersectionVolumeGuide(SC.random_scene(3, 4), "cpu", cfgs, 2)
    mc = ref_metrics.MetricsCalculator(guide)
    trs = trajectories()
    dt = 0.1
    jl, el, js, es, ee = [], [], [], [], []
    for tr in trs:
        a, b = mc.path_length_metric(tr)
        jl.append(a)
        el.append(b)
        sj, se = mc.smoothness_metric(tr, dt)
        # the reference's sparc returns (sal, (f, Mf), (f_sel, Mf_sel)); the all-zero profile returns (0, None, None)
        js.append(float(sj[0]))
        es.append(float(se[0]))
        jt = guide.rearrange_joints(torch.tensor(tr, dtype=torch.float32).unsqueeze(0))
        ee.append(guide.get_end_effector_transform(jt)[0, :, :3, 3].numpy())
    # the vendored third-party SPARC on the same speed profiles (it is what lib/metrics.py restates)
    tp = []
    for tr in trs:
        v = np.linalg.norm(np.diff(tr.T, axis=0) / dt, axis=1)
        tp.append(0.0 if np.allclose(v, 0) else float(ref_sparc(v, 1.0 / dt)[0]))
    print("  joint path length  ", np.round(jl, 4))
    print("  ee path length     ", np.round(el, 4))
    print("  joint SPARC        ", np.round(js, 4))
    print("  ee SPARC           ", np.round(es, 4))
    print("  third_party sparc == lib.metrics sparc (joint):", np.allclose(tp, js, atol=1e-12))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "g13_metrics.npz"), trajectories=trs, dt=dt, joint_path_length=np.array(jl), ee_path_length=np.array(el),
                        joint_sparc=np.array(js), ee_sparc=np.array(es), ee_positions=np.array(ee), third_party_joint_sparc=np.array(tp))
    print("  wrote g13_metrics.npz")

    # ---- G14: checkpoint written by the reference's own TemporalUNet.save()
    import diffusion.models.temporalunet as ref_unet  # noqa: E402

    tmp = tempfile.mkdtemp(prefix="edmp_ckpt_")
    name = os.path.join(tmp, "TemporalUNetModel255_N50")
    torch.manual_seed(21)
    net = ref_unet.TemporalUNet(model_name=name, input_dim=7, time_dim=32, dims=TINY_DIMS, device="cpu")
    net.save()
    files = sorted(os.listdir(name))
    print("  reference wrote:", files)
    x = torch.randn(3, 7, 50, generator=torch.Generator().manual_seed(5))
    t = torch.tensor([77.0])
    net.train(False)
    with torch.no_grad():
        y = net(x, t).numpy()
    dst = os.path.join(OUT, "g14_ref_checkpoint")
    shutil.rmtree(dst, ignore_errors=True)
    os.makedirs(dst)
    shutil.copy(os.path.join(name, "weights_latest.pt"), os.path.join(dst, "weights_latest.pt"))
    np.savez_compressed(os.path.join(dst, "forward.npz"), x=x.numpy(), t=t.numpy(), y=y, dims=np.array(TINY_DIMS))
    print(f"  wrote g14_ref_checkpoint/ ({os.path.getsize(os.path.join(dst, 'weights_latest.pt')) / 1024:.0f} KiB state dict written by the reference)")
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
