#!/usr/bin/env python
"""Golden vectors G15 for the link-mesh extent reader: runs the UNMODIFIED reference
IntersectionVolumeGuide.define_link_information (lib/guide.py:243-282) in this container on nine GENERATED Wavefront .obj
files that are not boxes (40-200 vertices each, 'vn' / 'vt' / 'f' / 'o' / '#' lines mixed in, tabs, repeated blanks,
exponent notation, indented lines), and writes tests/golden/g15_link_meshes.npz = {the .obj texts this script generated,
the reference's link_dimensions (9,3) f32 and link_vertices (9,4,8) f32}.  Needs /root/reference (build container only).

    python oracle/gen_golden_mesh.py"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import edmp_oracle as O  # noqa: E402
from oracle import ref_harness  # noqa: E402


def mesh_text(rs, k):
    """a vertex cloud inside a random box, written the ways .obj writers do"""
    n = int(rs.randint(40, 200))
    c = rs.uniform(-0.05, 0.05, 3)
    h = rs.uniform(0.01, 0.2, 3)
    v = c + rs.uniform(-1, 1, (n, 3)) * h
    out = [f"# generated mesh {k}", "o part", "mtllib none.mtl"]
    for i, p in enumerate(v):
        style = i % 5
        if style == 0:
            out.append(f"v {p[0]:.6f} {p[1]:.6f} {p[2]:.6f}")
        elif style == 1:
            out.append(f"v  {p[0]:.9e}\t{p[1]:.9e}   {p[2]:.9e}")
        elif style == 2:
            out.append(f"   v {float(p[0])!r} {float(p[1])!r} {float(p[2])!r} 1.0")  # indented, with the optional w coordinate
        elif style == 3:
            out.append(f"v {p[0]:.17g} {p[1]:.17g} {p[2]:.17g}  ")
        else:
            out.append(f"v\t{p[0]:.4f} {p[1]:.4f} {p[2]:.4f}")  # 'v<TAB>': NOT a vertex for the reference ('v ' prefix only)
        if i % 7 == 0:
            out.append(f"vn {rs.uniform(-9, 9):.4f} {rs.uniform(-9, 9):.4f} {rs.uniform(-9, 9):.4f}")  # must not widen the box
        if i % 11 == 0:
            out.append(f"vt {rs.uniform(-9, 9):.4f} {rs.uniform(-9, 9):.4f}")
    for i in range(1, n - 2, 3):
        out.append(f"f {i}//{i} {i + 1}//{i + 1} {i + 2}//{i + 2}")
    return "\n".join(out) + "\n"


def main():
    refd, refg = ref_harness.install(O.PLACEHOLDER_LINK_EXTENTS)
    import pybullet_data  # the harness's stand-in: getDataPath() -> a temp dir

    rs = np.random.RandomState(15)
    texts = [mesh_text(rs, k) for k in range(9)]
    datadir = tempfile.mkdtemp(prefix="edmp_g15_")
    mesh = os.path.join(datadir, "franka_panda", "meshes", "collision")
    os.makedirs(mesh)
    for name, t in zip(ref_harness.LINK_NAMES, texts):
        with open(os.path.join(mesh, name + ".obj"), "w") as f:
            f.write(t)
    with open(os.path.join(mesh, "README.txt"), "w") as f:  # a non-.obj file in the folder is skipped (lib/guide.py:256)
        f.write("v 100 100 100\n")
    pybullet_data.getDataPath = lambda: datadir
    import torch

    scene = np.array([[0.5, 0.0, 0.3, 0, 0, 0, 1, 0.2, 0.2, 0.2]])
    cfgs = {"batch_size_per_guide": 1, "total_batch_size": 1, "clearance": np.zeros((1, 255)), "expansion": np.zeros((1, 255)), "guidance_method": np.zeros(1),
            "grad_norm": np.zeros(1), "guidance_schedule": np.ones((1, 255)), "volume_trust_region": np.zeros(1)}
    g = refg.IntersectionVolumeGuide(scene, torch.device("cpu"), cfgs, 1)
    dims = g.link_dimensions.numpy()
    verts = g.link_vertices.numpy()
    print("reference link_dimensions:\n", dims)
    path = os.path.join(ROOT, "tests", "golden", "g15_link_meshes.npz")
    np.savez_compressed(path, obj_texts=np.array(texts), link_names=np.array(ref_harness.LINK_NAMES), link_dimensions=dims, link_vertices=verts)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
