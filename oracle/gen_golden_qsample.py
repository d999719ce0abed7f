#!/usr/bin/env python
"""Golden vectors G12 for the forward process: runs the UNMODIFIED reference Diffusion.q_sample /
q_sample_from_x0 / generate_q_sample (diffusion/diffusion.py:52-105, 201-251) in this container, asserts the oracle
restatement reproduces them bit for bit, and writes tests/golden/g12_qsample.npz.  Needs /root/reference (build
container only); the committed fixture is what travels.

    python oracle/gen_golden_qsample.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import edmp_oracle as O  # noqa: E402
from oracle import ref_harness  # noqa: E402

T = 255


def same(name, a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    assert np.array_equal(a, b), (name, float(np.abs(a - b).max()))
    print(f"  {name}: bit-exact {a.shape}")


def main():
    refd, _ = ref_harness.install(O.PLACEHOLDER_LINK_EXTENTS)
    dif = refd.Diffusion(T=T, device="cpu")
    beta, alpha, alpha_bar = O.schedule(T)
    rs = np.random.RandomState(12)
    x0 = rs.uniform(-2.5, 2.5, (6, 7, 50))
    out = {"x0": x0}

    # generate_q_sample, random timesteps, conditioned
    np.random.seed(77)
    X, Y, ts, means, vars_ = dif.generate_q_sample(x0.copy(), return_type="numpy")
    np.random.seed(77)
    oX, oY, ots, omeans, ovars = O.generate_q_sample(x0.copy(), T, alpha_bar)
    for n, a, b in (("X", oX, X), ("Y", oY, Y), ("time_steps", ots, ts), ("means", omeans, means), ("vars", ovars, vars_)):
        same("generate_q_sample." + n, a, b)
    out.update(gq_seed=np.array(77), gq_X=X, gq_Y=Y, gq_t=ts, gq_means=means, gq_vars=vars_)

    # given timesteps (first, last, repeated), unconditioned
    tt = np.array([1, 255, 128, 128, 2, 254])
    np.random.seed(78)
    X2, Y2, ts2, means2, vars2 = dif.generate_q_sample(x0.copy(), time_steps=tt.copy(), condition=False, return_type="numpy")
    np.random.seed(78)
    oX2, oY2, _, omeans2, ovars2 = O.generate_q_sample(x0.copy(), T, alpha_bar, time_steps=tt.copy(), condition=False)
    for n, a, b in (("X", oX2, X2), ("Y", oY2, Y2), ("means", omeans2, means2), ("vars", ovars2, vars2)):
        same("generate_q_sample[given t, no cond]." + n, a, b)
    out.update(gq2_seed=np.array(78), gq2_t=tt, gq2_X=X2, gq2_Y=Y2, gq2_means=means2, gq2_vars=vars2)

    # the float32 tensors of return_type="tensor"
    np.random.seed(79)
    Xt, Yt, tst, _, _ = dif.generate_q_sample(x0.copy(), return_type="tensor")
    out.update(gq3_seed=np.array(79), gq3_X32=Xt.numpy(), gq3_Y32=Yt.numpy(), gq3_t32=tst.numpy())

    # single forward step q(x_t | x_{t-1}) with explicit eps, and with the internal draw
    eps = rs.standard_normal((6, 7, 50))
    xs, ms, vs = dif.q_sample(x0.copy(), tt, eps)
    oxs, oms, ovs = O.q_sample(x0.copy(), tt, eps, alpha)
    for n, a, b in (("xt", oxs, xs), ("mean", oms, ms), ("var", ovs, vs)):
        same("q_sample." + n, a, b)
    out.update(qs_t=tt, qs_eps=eps, qs_xt=xs, qs_mean=ms, qs_var=vs)
    np.random.seed(80)
    xs2, _, _ = dif.q_sample_from_x0(x0.copy(), tt)
    np.random.seed(80)
    e2 = np.random.standard_normal((6, 350)).reshape(6, 7, 50)
    same("q_sample_from_x0[internal draw]", O.q_sample_from_x0(x0.copy(), tt, e2, alpha_bar)[0], xs2)
    out.update(q0_seed=np.array(80), q0_xt=xs2)

    path = os.path.join(ROOT, "tests", "golden", "g12_qsample.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
