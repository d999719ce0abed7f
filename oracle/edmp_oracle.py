"""CPU oracle for the EDMP guided reverse-diffusion sampler — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A restatement (NumPy float64 + torch-CPU float32, exactly the number types the reference uses) of the hot path
named by BASELINE.json: ``Diffusion.denoise_guided`` and everything it calls.  Every function cites the reference
lines (relative to /root/reference) whose behaviour it restates.  Parity status: PINNED — ``oracle/gen_golden.py``
imports the unmodified reference in the build container, asserts this file reproduces it, and commits the resulting
vectors under ``tests/golden/`` (the reference itself ships no tests / golden vectors).
Third-party inputs that the reference does not pin (link-mesh extents from pybullet_data, IK goals, pybullet
success) are explicit *inputs* here: "parity unpinned" for those, see DESIGN.md.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------------------------
# a1  variance schedule                                                       diffusion/diffusion.py:10-20, 37-49
# --------------------------------------------------------------------------------------------------------------


def schedule(T: int, variance_thresh: float = 0.02):
    """beta_t = linspace(0, thresh, T+1)[1:], alpha = 1-beta, alpha_bar_t = prod(alpha[:t]).  (diffusion.py:13-16,47)"""
    beta = np.linspace(0, variance_thresh, T + 1)[1:]
    alpha = 1 - beta
    alpha_bar = np.array([np.prod(alpha[:t]) for t in np.arange(T + 1)[1:]])
    return beta, alpha, alpha_bar


# --------------------------------------------------------------------------------------------------------------
# a2  posterior step                                                                  diffusion/diffusion.py:116-135
# --------------------------------------------------------------------------------------------------------------


def p_sample_using_posterior(xt, t, eps, z, beta, alpha, alpha_bar):
    """x <- (x - (1-a_t)/sqrt(1-abar_t) * eps)/sqrt(a_t) + beta_t * z      (diffusion.py:129-133)

    ``z`` is the (B,C,N) float64 standard-normal draw the reference makes at diffusion.py:126 (drawn by the caller
    so the RNG stream is explicit).  Quirk Q3 (diffusion.py:127, NumPy-1.x semantics of ``np.where(t == 1)`` on a
    Python int): at t == 1 only batch row 0 of z is zeroed; otherwise a no-op.  Quirk Q1: noise scale is beta_t.
    """
    z = np.array(z, dtype=np.float64, copy=True)
    if t == 1:
        z[0, :, :] = 0
    a = alpha[t - 1]
    ab = alpha_bar[t - 1]
    b = beta[t - 1]
    return ((xt - ((1 - a) / np.sqrt(1 - ab)) * eps) / np.sqrt(a) + b * z).copy()


# --------------------------------------------------------------------------------------------------------------
# f4  forward process (training-side data generation)        diffusion/diffusion.py:52-105, 201-251
# --------------------------------------------------------------------------------------------------------------


def q_sample(x, t, eps, alpha):
    """one forward step q(x_t | x_{t-1}): (xt, mean, var) of diffusion.py:72-76; ``t`` an int array (b,)."""
    t = np.asarray(t)
    a = alpha[t - 1, np.newaxis, np.newaxis]
    xt = np.sqrt(a) * x + np.sqrt(1 - a) * eps
    return xt, np.sqrt(a) * x, np.sqrt(1 - alpha[t - 1])


def q_sample_from_x0(x0, t, eps, alpha_bar):
    """q(x_t | x_0): (xt, mean, var) of diffusion.py:100-104 (var keeps the (b,1,1) shape the reference returns)."""
    t = np.asarray(t)
    ab = alpha_bar[t - 1, np.newaxis, np.newaxis]
    xt = np.sqrt(ab) * x0 + np.sqrt(1 - ab) * eps
    return xt, np.sqrt(ab) * x0, np.sqrt(1 - ab)


def generate_q_sample(x0, T, alpha_bar, time_steps=None, condition=True):
    """diffusion.py:201-251 with return_type="numpy": draws (global NumPy RNG, reference order) the timesteps
    ``randint(1, T+1, (b,))`` unless given, then eps = multivariate_normal(0, I_n, (b, c)) (== standard_normal((b,c,n))
    for an identity covariance), diffuses, and pins the first / last waypoint to x0 when ``condition``.
    Returns (X, Y, time_steps, means, vars)."""
    b, c, n = x0.shape
    if time_steps is None:
        time_steps = np.random.randint(1, T + 1, size=(b,))
    eps = np.random.standard_normal((b, c, n))
    xt, means, vars_ = q_sample_from_x0(x0, time_steps, eps, alpha_bar)
    if condition:
        xt[:, :, 0] = x0[:, :, 0].copy()
        xt[:, :, -1] = x0[:, :, -1].copy()
    return xt.copy(), eps.copy(), time_steps, means, vars_


# --------------------------------------------------------------------------------------------------------------
# a3  joint clip                                                                      diffusion/diffusion.py:280-298
# --------------------------------------------------------------------------------------------------------------

JOINT_LOWER_DEG = np.array([-166.0, -101.0, -166.0, -176.0, -166.0, -1.0, -166.0])
JOINT_UPPER_DEG = np.array([166.0, 101.0, 166.0, -4.0, 166.0, 215.0, 166.0])


def joint_limits():
    return JOINT_LOWER_DEG * (np.pi / 180), JOINT_UPPER_DEG * (np.pi / 180)


def clip_joints(joints):
    lo, hi = joint_limits()
    return np.clip(joints, lo[np.newaxis, :, np.newaxis], hi[np.newaxis, :, np.newaxis])


# --------------------------------------------------------------------------------------------------------------
# a16 per-row guide parameter arrays                                                       infer_serial.py:56-91
# --------------------------------------------------------------------------------------------------------------


def build_guide_cfgs(guide_dicts, batch_size_per_guide: int, T: int):
    """``guide_dicts`` = list of parsed guide YAML dicts (the reference loads guides/cfgs/guide<N>.yaml).

    clearance rows = linspace(r0, r1, T); expansion segments isr1, isr2, isr3 written IN THAT ORDER (later
    overwrite earlier); method 1 for 'sv'; schedule 1.4 + arange(T)/T if 'varying' else scale_val.
    """
    G = len(guide_dicts)
    bpg = batch_size_per_guide
    B = int(G * bpg)
    cfgs = {
        "batch_size_per_guide": bpg,
        "total_batch_size": B,
        "clearance": np.zeros((B, T)),
        "expansion": np.zeros((B, T)),
        "guidance_method": np.zeros((B,)),
        "grad_norm": np.zeros((B,)),
        "guidance_schedule": np.zeros((B, T)),
        "volume_trust_region": np.zeros((B,)),
    }
    for i, g in enumerate(guide_dicts):
        h = g["hyperparameters"]
        rows = slice(i * bpg, (i + 1) * bpg)
        r = h["obstacle_clearance"]["range"]
        cfgs["clearance"][rows, :] = np.linspace(r[0], r[1], T)
        oe = h["obstacle_expansion"]
        for k in ("1", "2", "3"):
            lo, hi = oe["isr" + k]
            v = oe["val" + k]
            cfgs["expansion"][rows, lo:hi] = np.linspace(v[0], v[1], num=abs(hi - lo))
        cfgs["guidance_method"][rows] = 1 if h["guidance_method"] == "sv" else 0
        cfgs["grad_norm"][rows] = 1 if h["grad_norm"] else 0
        gs = h["guidance_schedule"]
        cfgs["guidance_schedule"][rows, :] = (1.4 + np.arange(T) / T) if gs["type"] == "varying" else gs["scale_val"]
        cfgs["volume_trust_region"][rows] = h["volume_trust_region"]
    return cfgs


# --------------------------------------------------------------------------------------------------------------
# a7-a9  robot model: DH chain, link boxes, static frames                      lib/guide.py:29-38, 203-241, 286-342
# --------------------------------------------------------------------------------------------------------------

PI = math.pi
# rows [a, d, alpha, theta0]; only the first 7 are on the hot path (lib/guide.py:29-38, 88)
STATIC_DH = [
    [0, 0.333, 0, 0],
    [0, 0, -PI / 2, 0],
    [0, 0.316, PI / 2, 0],
    [0.0825, 0, PI / 2, 0],
    [-0.0825, 0.384, -PI / 2, 0],
    [0, 0, PI / 2, 0],
    [0.088, 0, PI / 2, 0],
]
LINK_FRAME = [0, 1, 2, 3, 4, 5, 6, 6, 6]  # lib/guide.py:93-94, 286 (0-based cumulative-transform index)
_C45, _S45 = 7.07106767e-01, 7.07106795e-01
STATIC_FRAMES = [  # lib/guide.py:289-340
    [[1, 0, 0, 8.71e-05], [0, 1, 0, -3.709035e-02], [0, 0, 1, -6.851545e-02], [0, 0, 0, 1]],
    [[1, 0, 0, -8.425e-05], [0, 1, 0, -6.93950016e-02], [0, 0, 1, 3.71961970e-02], [0, 0, 0, 1]],
    [[1, 0, 0, 0.0414576], [0, 1, 0, 0.0281429], [0, 0, 1, -0.03293086], [0, 0, 0, 1]],
    [[1, 0, 0, -4.12337575e-02], [0, 1, 0, 3.44296512e-02], [0, 0, 1, 2.79226985e-02], [0, 0, 0, 1]],
    [[1, 0, 0, 3.3450000e-05], [0, 1, 0, 3.7388050e-02], [0, 0, 1, -1.0619285e-01], [0, 0, 0, 1]],
    [[1, 0, 0, 4.21935000e-02], [0, 1, 0, 1.52195003e-02], [0, 0, 1, 6.07699933e-03], [0, 0, 0, 1]],
    [[1, 0, 0, 1.86357500e-02], [0, 1, 0, 1.85788569e-02], [0, 0, 1, 7.94137484e-02], [0, 0, 0, 1]],
    [[_C45, _S45, 0, -1.26717073e-03], [-_S45, _C45, 0, -1.25294673e-03], [0, 0, 1, 1.27018693e-01], [0, 0, 0, 1]],
    [[_C45, _S45, 0, 9.29352476e-03], [-_S45, _C45, 0, 9.28272434e-03], [0, 0, 1, 1.92390375e-01], [0, 0, 0, 1]],
]
# sign pattern of the 8 box corners, column order of the (4,8) vertex matrix (lib/guide.py:210-235, 170-195)
VERT_SX = [-1, 1, 1, -1, -1, 1, 1, -1]
VERT_SY = [-1, -1, 1, 1, -1, -1, 1, 1]
VERT_SZ = [-1, -1, -1, -1, 1, 1, 1, 1]

# PLACEHOLDER link-box extents (l, b, h) for link1..link7, hand, finger.  The reference derives these from
# pybullet_data's Franka collision meshes (lib/guide.py:245-282), which are absent here: PARITY UNPINNED for these
# numbers.  They are data; oracle, golden generator and product use the same table.  The finger's y extent is
# given BEFORE the reference's x4 (lib/guide.py:278-279), which `link_dimensions_effective` applies.
PLACEHOLDER_LINK_EXTENTS = np.array(
    [
        [0.110, 0.174, 0.260],
        [0.110, 0.260, 0.175],
        [0.180, 0.170, 0.190],
        [0.180, 0.175, 0.170],
        [0.110, 0.190, 0.360],
        [0.185, 0.140, 0.115],
        [0.110, 0.110, 0.095],
        [0.065, 0.205, 0.095],
        [0.022, 0.016, 0.055],
    ],
    dtype=np.float64,
)


def link_dimensions_effective(mesh_extents):
    d = np.array(mesh_extents, dtype=np.float64, copy=True)
    d[-1, 1] *= 4  # lib/guide.py:278-279
    return torch.tensor(d, dtype=torch.float32)


def box_vertices(dims: torch.Tensor) -> torch.Tensor:
    """(..., 3) extents -> (..., 4, 8) homogeneous corners in the reference's column order."""
    l, b, h = dims[..., 0], dims[..., 1], dims[..., 2]
    v = torch.zeros(*dims.shape[:-1], 4, 8, dtype=torch.float32)
    for i in range(8):
        v[..., 0, i] = (l / 2) * VERT_SX[i]
        v[..., 1, i] = (b / 2) * VERT_SY[i]
        v[..., 2, i] = (h / 2) * VERT_SZ[i]
    v[..., 3, :] = 1.0
    return v


def get_tf_mat(dh):
    """modified-DH transform, lib/guide.py:45-72.  dh: (b, n, 4) = [a, d, alpha, q]."""
    a, d, al, q = dh[:, :, 0], dh[:, :, 1], dh[:, :, 2], dh[:, :, 3]
    tf = torch.zeros(dh.shape[0], dh.shape[1], 4, 4)
    tf[:, :, 0, 0] = torch.cos(q)
    tf[:, :, 0, 1] = -torch.sin(q)
    tf[:, :, 0, 3] = a
    tf[:, :, 1, 0] = torch.sin(q) * torch.cos(al)
    tf[:, :, 1, 1] = torch.cos(q) * torch.cos(al)
    tf[:, :, 1, 2] = -torch.sin(al)
    tf[:, :, 1, 3] = -torch.sin(al) * d
    tf[:, :, 2, 0] = torch.sin(q) * torch.sin(al)
    tf[:, :, 2, 1] = torch.cos(q) * torch.sin(al)
    tf[:, :, 2, 2] = torch.cos(al)
    tf[:, :, 2, 3] = torch.cos(al) * d
    tf[:, :, 3, 3] = 1
    return tf


def forward_kinematics(joints):
    """joints (b, n, 7) f32 -> (b, n, 9, 4, 4) cumulative joint frames, links 7,8 ride frame 6.  lib/guide.py:74-98"""
    b, n = joints.shape[0], joints.shape[1]
    dh = torch.tensor(STATIC_DH, dtype=torch.float32).unsqueeze(0).unsqueeze(0).repeat(b, n, 1, 1)
    dh[:, :, :7, 3] = joints
    fk = torch.zeros(b, n, 9, 4, 4, dtype=torch.float32)
    T = torch.eye(4).unsqueeze(0).unsqueeze(0).repeat(b, n, 1, 1)
    for i in range(7):
        T = torch.matmul(T, get_tf_mat(dh[:, :, i, :]))
        if i == 6:
            fk[:, :, i:, :, :] = T.unsqueeze(2)
        else:
            fk[:, :, i, :, :] = T
    return fk


def get_link_transform(joints):
    """lib/guide.py:344-352"""
    sf = torch.tensor(STATIC_FRAMES, dtype=torch.float32)
    return forward_kinematics(joints) @ sf.unsqueeze(0).unsqueeze(0)


def link_aabbs(joints, link_dims):
    """(b, n, 7) -> link_min, link_max (b, n, 9, 3).  lib/guide.py:361-375"""
    lt = get_link_transform(joints)
    lv = lt @ box_vertices(link_dims).unsqueeze(0).unsqueeze(0)
    lv = lv[:, :, :, :3, :]
    return torch.min(lv, dim=-1)[0], torch.max(lv, dim=-1)[0]


# --------------------------------------------------------------------------------------------------------------
# a10 obstacle AABBs                                                                       lib/guide.py:118-201
# --------------------------------------------------------------------------------------------------------------


def quat_xyzw_to_matrix(q):
    """What scipy.spatial.transform.Rotation.from_quat(q).as_matrix() computes (scalar-last, normalised, f64).
    Call site: lib/guide.py:143.  (scipy is third-party; pinned against scipy itself in tests.)"""
    q = np.asarray(q, dtype=np.float64)
    x, y, z, w = q / np.linalg.norm(q)
    x2, y2, z2, w2 = x * x, y * y, z * z, w * w
    xy, zw, xz, yw, yz, xw = x * y, z * w, x * z, y * w, y * z, x * w
    return np.array(
        [
            [x2 - y2 - z2 + w2, 2 * (xy - zw), 2 * (xz + yw)],
            [2 * (xy + zw), -x2 + y2 - z2 + w2, 2 * (yz - xw)],
            [2 * (xz - yw), 2 * (yz + xw), -x2 - y2 + z2 + w2],
        ]
    )


def define_obstacles(obstacle_config, clearance, expansion, t: int, b: int):
    """-> obs_min, obs_max (b, no, 3) f32.  lib/guide.py:118-158.

    sizes <- max(sizes, expansion[:, t-1]) + clearance[:, t-1] for t != 0 (f64), then f32 corners, f32 matmul with
    the f32 obstacle transform, min/max over the 8 corners."""
    oc = np.array(obstacle_config, dtype=np.float64)
    sizes = np.repeat(oc[np.newaxis, :, 7:], b, axis=0)
    if t != 0:
        sizes = np.maximum(sizes, expansion[:, t - 1, np.newaxis, np.newaxis])
        sizes = sizes + clearance[:, t - 1, np.newaxis, np.newaxis]
    sv = box_vertices(torch.tensor(sizes, dtype=torch.float32))
    tf = np.zeros((oc.shape[0], 4, 4))
    for i in range(oc.shape[0]):
        tf[i, :3, :3] = quat_xyzw_to_matrix(oc[i, 3:7])
        tf[i, :3, -1] = oc[i, :3]
    tf[:, -1, -1] = 1.0
    tf = torch.tensor(np.repeat(tf[np.newaxis], b, axis=0), dtype=torch.float32)
    ov = torch.matmul(tf, sv)
    return torch.min(ov, dim=-1)[0][:, :, :-1], torch.max(ov, dim=-1)[0][:, :, :-1]


# --------------------------------------------------------------------------------------------------------------
# a11/a12 costs                                                                   lib/guide.py:354-395, 473-537
# --------------------------------------------------------------------------------------------------------------


def _overlap_volumes(lmin, lmax, omin, omax):
    """lmin/lmax (b, n, 9, 3); omin/omax (b, no, 3) -> (b, n, 9*no) with pair index link*no + obs."""
    b, n, nl = lmin.shape[0], lmin.shape[1], lmin.shape[2]
    no = omin.shape[1]
    elmin = lmin.unsqueeze(-2).repeat(1, 1, 1, no, 1).view(b, n, no * nl, 3)
    elmax = lmax.unsqueeze(-2).repeat(1, 1, 1, no, 1).view(b, n, no * nl, 3)
    eomin = omin.unsqueeze(1).unsqueeze(1).repeat(1, n, nl, 1, 1).view(b, n, no * nl, 3)
    eomax = omax.unsqueeze(1).unsqueeze(1).repeat(1, n, nl, 1, 1).view(b, n, no * nl, 3)
    lengths = torch.min(elmax, eomax) - torch.max(elmin, eomin)
    return torch.prod(torch.clamp(lengths, min=0), dim=-1)


def cost_iv(joint_input, omin, omax, link_dims):
    """joint_input (b, 7, n) f32 tensor -> volumes (b, n, 9*no).  lib/guide.py:354-395"""
    joints = joint_input.permute(0, 2, 1)
    lmin, lmax = link_aabbs(joints, link_dims)
    return _overlap_volumes(lmin, lmax, omin, omax)


def cost_sv(joint_input, start, goal, omin, omax, link_dims):
    """swept volume: pad with start/goal, AABB of consecutive waypoints' AABBs.  lib/guide.py:473-537"""
    joints = joint_input.permute(0, 2, 1)
    b, n = joints.shape[0], joints.shape[1]
    traj = torch.zeros(b, n + 2, 7)
    traj[:, 0, :] = start.unsqueeze(0).repeat(b, 1)
    traj[:, -1, :] = goal.unsqueeze(0).repeat(b, 1)
    traj[:, 1:-1, :] = joints
    lmin, lmax = link_aabbs(traj, link_dims)
    smin = torch.min(lmin[:, :-1], lmin[:, 1:])
    smax = torch.max(lmax[:, :-1], lmax[:, 1:])
    return _overlap_volumes(smin, smax, omin, omax)


# --------------------------------------------------------------------------------------------------------------
# a13 gradient, a14 best trajectory                                               lib/guide.py:597-635, 637-653
# --------------------------------------------------------------------------------------------------------------


class GuideOracle:
    """Restates IntersectionVolumeGuide (lib/guide.py:11-653) for the hot-path methods."""

    def __init__(self, obstacle_config, guide_cfgs, batch_size, link_mesh_extents=None):
        self.obstacle_config = np.array(obstacle_config)
        self.guide_cfgs = guide_cfgs
        self.batch_size = batch_size
        ext = PLACEHOLDER_LINK_EXTENTS if link_mesh_extents is None else link_mesh_extents
        self.link_dims = link_dimensions_effective(ext)

    def obstacles(self, t, batch_size=None):
        b = self.batch_size if batch_size is None else batch_size
        return define_obstacles(self.obstacle_config, self.guide_cfgs["clearance"], self.guide_cfgs["expansion"], t, b)

    def cost(self, joint_tensor, t, batch_size=None):
        omin, omax = self.obstacles(t, batch_size)
        return cost_iv(torch.as_tensor(joint_tensor, dtype=torch.float32), omin, omax, self.link_dims)

    def swept_volume_cost(self, joint_tensor, start, goal, t, batch_size=None):
        omin, omax = self.obstacles(t, batch_size)
        return cost_sv(
            torch.as_tensor(joint_tensor, dtype=torch.float32),
            torch.as_tensor(start, dtype=torch.float32),
            torch.as_tensor(goal, dtype=torch.float32),
            omin,
            omax,
            self.link_dims,
        )

    def raw_gradient(self, joint_input, start, goal, t):
        """f32 autograd gradient before grad_norm mixing (lib/guide.py:599-623)."""
        jt = torch.tensor(joint_input, dtype=torch.float32)
        jt.requires_grad = True
        start = torch.tensor(start, dtype=torch.float32)
        goal = torch.tensor(goal, dtype=torch.float32)
        b = self.batch_size
        m = torch.tensor(self.guide_cfgs["guidance_method"], dtype=torch.float32).view(b, 1, 1)
        omin, omax = self.obstacles(t)
        cost = torch.sum((1 - m) * cost_iv(jt, omin, omax, self.link_dims)) + torch.sum(
            m * cost_sv(jt, start, goal, omin, omax, self.link_dims)
        )
        cost.backward()
        return jt.grad.cpu().numpy()

    def get_gradient(self, joint_input, start, goal, t):
        """lib/guide.py:597-635 incl. the whole-batch Frobenius norm (Q5) and 0*NaN poisoning (Q7)."""
        g = self.raw_gradient(joint_input, start, goal, t)
        gn = self.guide_cfgs["grad_norm"][:, np.newaxis, np.newaxis]
        with np.errstate(divide="ignore", invalid="ignore"):
            return (1 - gn) * g + gn * (g / np.linalg.norm(g))

    def row_swept_volumes(self, start, goal, trajectories):
        jt = torch.tensor(trajectories[:, :, 1:-1], dtype=torch.float32)
        omin, omax = self.obstacles(0)
        v = cost_sv(
            jt, torch.tensor(start, dtype=torch.float32), torch.tensor(goal, dtype=torch.float32), omin, omax, self.link_dims
        )
        return torch.sum(v, dim=(1, 2))

    def choose_best_trajectory(self, start, goal, trajectories):
        """argmin of the t=0 swept volume, first index on ties.  lib/guide.py:637-653"""
        vols = self.row_swept_volumes(start, goal, trajectories)
        return trajectories[torch.argmin(vols)]


# --------------------------------------------------------------------------------------------------------------
# a5/a6 TemporalUNet forward                       diffusion/models/temporalunet.py:47-76, blocks.py:13-260
# --------------------------------------------------------------------------------------------------------------


def sinusoidal_pos_emb(t: torch.Tensor, dim: int):
    """blocks.py:46-54"""
    half = dim // 2
    e = np.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half) * -e)
    e = t[:, None] * e[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def _conv_block(sd, p, x):
    """Conv1d(k, pad k//2) -> GroupNorm(8) -> Mish.  blocks.py:22-28"""
    w = sd[p + ".block.0.weight"]
    x = F.conv1d(x, w, sd[p + ".block.0.bias"], padding=w.shape[-1] // 2)
    x = F.group_norm(x, 8, sd[p + ".block.2.weight"], sd[p + ".block.2.bias"], eps=1e-5)
    return F.mish(x)


def _rcb(sd, p, x, temb):
    """ResidualConvolutionBlock.forward, blocks.py:154-166"""
    tb = F.linear(F.mish(temb), sd[p + ".time_mlp.time_mlp.1.weight"], sd[p + ".time_mlp.time_mlp.1.bias"])
    out = _conv_block(sd, p + ".blocks.0", x) + tb[:, :, None]
    out = _conv_block(sd, p + ".blocks.1", out)
    if (p + ".residual_conv.weight") in sd:
        res = F.conv1d(x, sd[p + ".residual_conv.weight"], sd[p + ".residual_conv.bias"])
    else:
        res = x
    return out + res


def unet_forward(sd, x: torch.Tensor, t: torch.Tensor, time_dim: int = 32, trace: dict | None = None):
    """sd: state dict (names as in SURVEY.md §8a 'U'); x (B, C, N) f32; t (1,) f32 -> (B, C, N) f32."""
    n_down = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("down_samplers."))
    n_up = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("up_samplers."))
    temb = sinusoidal_pos_emb(t, time_dim)
    temb = F.linear(temb, sd["time_embedding.time_mlp.1.weight"], sd["time_embedding.time_mlp.1.bias"])
    temb = F.linear(F.mish(temb), sd["time_embedding.time_mlp.3.weight"], sd["time_embedding.time_mlp.3.bias"])
    hs = []
    for i in range(n_down):
        p = f"down_samplers.{i}.down"
        x = _rcb(sd, p + ".0", x, temb)
        x = _rcb(sd, p + ".1", x, temb)
        hs.append(x)
        if (p + ".3.weight") in sd:
            x = F.conv1d(x, sd[p + ".3.weight"], sd[p + ".3.bias"], stride=2, padding=1)
        if trace is not None:
            trace[f"down{i}"] = x.clone()
    x = _rcb(sd, "middle_block.middle.0", x, temb)
    x = _rcb(sd, "middle_block.middle.2", x, temb)
    if trace is not None:
        trace["mid"] = x.clone()
    for i in range(n_up):
        p = f"up_samplers.{i}.up"
        x = torch.cat([x, hs.pop()], dim=1)
        x = _rcb(sd, p + ".0", x, temb)
        x = _rcb(sd, p + ".1", x, temb)
        x = F.conv_transpose1d(x, sd[p + ".3.weight"], sd[p + ".3.bias"], stride=2, padding=1)
        if x.shape[2] in (8, 14, 26):  # temporalunet.py:70-71
            x = x[:, :, : x.shape[2] - 1]
        if trace is not None:
            trace[f"up{i}"] = x.clone()
    x = _conv_block(sd, "final_conv.0", x)
    return F.conv1d(x, sd["final_conv.1.weight"], sd["final_conv.1.bias"])


class UNetOracle:
    def __init__(self, state_dict, time_dim=32):
        self.sd = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in state_dict.items()}
        self.time_dim = time_dim

    def __call__(self, x, t):
        with torch.no_grad():
            return unet_forward(self.sd, x, t, self.time_dim)

    def train(self, flag):
        return self


# --------------------------------------------------------------------------------------------------------------
# a4/a15 the guided loop                                                           diffusion/diffusion.py:300-356
# --------------------------------------------------------------------------------------------------------------


def denoise_step(model, guide, X, z, t, guidance_schedule, start, goal, sched, condition=True, full=True):
    """One iteration of the reverse loop (diffusion.py:314-349) from state X (B,C,N) f64 with the step's draw z:
    dict(x_in, eps, x_post, grad, x_out).  Teacher-forced tests call it on arbitrary states."""
    beta, alpha, alpha_bar = sched
    x_in = X.copy() if full else None
    eps = model(torch.tensor(X, dtype=torch.float32), torch.tensor([t], dtype=torch.float32)).numpy(force=True)
    X = p_sample_using_posterior(X, t, eps, z, beta, alpha, alpha_bar)
    x_post = X.copy() if full else None
    grad = None
    if (t % 2) < 1 and t >= 5:
        grad = guide.get_gradient(clip_joints(X[:, :, 1:-1]), start[:], goal[:], t)
        X[:, :, 1:-1] = X[:, :, 1:-1] - guidance_schedule[:, t - 1, np.newaxis, np.newaxis] * grad
    if condition:  # diffusion.py:347-349
        X[:, :, 0] = start[:]
        X[:, :, -1] = goal[:]
    return dict(x_in=x_in, eps=eps, x_post=x_post, grad=grad, x_out=X if not full else X.copy())


def denoise_guided(
    model, guide, T, traj_len, num_channels, guidance_schedule, batch_size, start, goal, noise=None, trace=None, t_stop=0, condition=True
):
    """Restates Diffusion.denoise_guided.  ``noise`` (T+1, B, C, N) f64: noise[0] is the initial draw
    (diffusion.py:303), noise[1 + (T - t)] the draw of step t (diffusion.py:126); if None it is drawn from the global
    NumPy RandomState in the reference's order.  ``trace`` (dict) receives per-step x_in, eps, x_post, grad, x_out.
    ``t_stop``: run steps T..t_stop+1 only (bounded CPU-baseline sample)."""
    beta, alpha, alpha_bar = schedule(T)
    if noise is None:
        noise = np.random.standard_normal((T + 1, batch_size, num_channels, traj_len))
    X = np.array(noise[0], dtype=np.float64, copy=True)
    if condition:  # diffusion.py:305-307
        X[:, :, 0] = start[:]
        X[:, :, -1] = goal[:]
    for t in range(T, t_stop, -1):
        st = denoise_step(model, guide, X, noise[1 + (T - t)], t, guidance_schedule, start, goal, (beta, alpha, alpha_bar), condition, full=trace is not None)
        X = st["x_out"]
        if trace is not None:
            trace[t] = st
    return X.copy()
