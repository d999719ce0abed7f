"""IntersectionVolumeGuide — host-side mirror of the reference guide object (lib/guide.py:11-653) whose arithmetic
runs in libedmp_hip.so (edmp_amd/csrc/guide.hip)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi, franka
from .runtime import get_context, ptr, new_slot_key


def row_classes(clearance: np.ndarray, expansion: np.ndarray):
    """Rows with identical (clearance[t], expansion[t]) schedules share one obstacle table: returns
    (row_class (B,) int32, class_clearance (G,T), class_expansion (G,T))."""
    B = clearance.shape[0]
    # a guide owns a contiguous block of rows (infer_serial.py:56-91), so consecutive rows are nearly always equal: compare every
    # row with its predecessor in one vectorised pass and key only the first row of each run (a Python loop over 1024 rows with two
    # tobytes() each cost 1.5 ms per scene - the guide object is rebuilt for every scene, infer_serial.py:112)
    key = np.ascontiguousarray(np.concatenate([clearance, expansion], axis=1)).view(np.uint64)  # bit patterns: -0.0 != 0.0, NaN == NaN
    starts = np.concatenate([[0], 1 + np.flatnonzero(np.any(key[1:] != key[:-1], axis=1))]) if B > 1 else np.array([0])
    keys = {}
    reps = []
    cls_of_run = np.empty(len(starts), dtype=np.int32)
    for j, b in enumerate(starts):
        k = key[b].tobytes()
        if k not in keys:
            keys[k] = len(reps)
            reps.append(int(b))
        cls_of_run[j] = keys[k]
    rc = np.repeat(cls_of_run, np.diff(np.concatenate([starts, [B]]))).astype(np.int32)
    return rc, np.ascontiguousarray(clearance[reps], dtype=np.float64), np.ascontiguousarray(expansion[reps], dtype=np.float64)


class IntersectionVolumeGuide:
    """Same constructor / method signatures as the reference:

        guide = IntersectionVolumeGuide(obstacle_config, device, guide_cfgs, batch_size)
        guide.cost(joint_tensor (n,7,L), t, batch_size=None)              -> (n, L, 9*no) f32 tensor
        guide.swept_volume_cost(joint_tensor, start, goal, t, batch_size=None) -> (n, L+1, 9*no)
        guide.get_gradient(joint_input (B,7,48) ndarray, start, goal, t)  -> (B,7,48) f64 ndarray
        guide.choose_best_trajectory(start, goal, trajectories (B,7,50))  -> (7,50)

    Link boxes: the reference measures the Franka collision meshes of pybullet_data every time a guide is built
    (lib/guide.py:245-282).  Here (franka.resolve_link_extents): ``link_mesh_extents`` (9,3) if given, else the meshes in
    ``mesh_dir``, else pybullet_data's directory when importable, else the placeholder table with one warning per process.
    """

    def __init__(self, obstacle_config, device, guide_cfgs, batch_size, *, link_mesh_extents=None, mesh_dir=None, obstacle_kinds=None):
        self.ctx = get_context(device)
        self.device = self.ctx.device
        self.guide_cfgs = guide_cfgs
        self.obstacle_config = np.ascontiguousarray(np.array(obstacle_config, dtype=np.float64))
        if self.obstacle_config.ndim != 2 or self.obstacle_config.shape[1] != 10:
            raise ValueError("obstacle_config must be (n_obstacles, 10) = [xyz, quat xyzw, dims]")
        self.batch_size = int(batch_size)
        self.T = int(np.asarray(guide_cfgs["clearance"]).shape[1])
        clr = np.asarray(guide_cfgs["clearance"], dtype=np.float64)
        exp = np.asarray(guide_cfgs["expansion"], dtype=np.float64)
        if clr.shape[0] != self.batch_size:
            raise ValueError(f"guide_cfgs rows ({clr.shape[0]}) != batch_size ({self.batch_size})")
        self.row_class, self._cls_clr, self._cls_exp = row_classes(clr, exp)
        self.link_mesh_extents = franka.resolve_link_extents(link_mesh_extents, mesh_dir)
        self._half = np.ascontiguousarray(franka.link_half_extents(self.link_mesh_extents))
        self.link_dimensions = torch.from_numpy(self._half * 2)
        self._dh = np.ascontiguousarray(franka.dh_table())
        self._sf = np.ascontiguousarray(franka.static_frames())
        self._sched = np.ascontiguousarray(np.asarray(guide_cfgs["guidance_schedule"], dtype=np.float64))
        self._rows_token = None
        self._slot = new_slot_key()
        # 0 cuboid / 1 cylinder per obstacle row: only the success check reads it (the guide itself sees cylinders as
        # (r, r, h) boxes, quirk Q9).  The reference's loader orders obstacle_config cuboids first, then cylinders
        # (datasets/load_test_dataset.py:141-149), so kinds = [0] * num_cuboids + [1] * num_cylinders there.
        self._kinds = None if obstacle_kinds is None else self._check_kinds(obstacle_kinds)
        self._bind()

    # ---- binding -----------------------------------------------------------------------------------------------
    def _bind(self):
        ctx = self.ctx
        if ctx.bound_guide is self:
            return
        # switch to this object's resident slot; the scene tables / row arrays are rebuilt only if the slot is empty
        have = ctx.lib.edmp_guide_slot(ctx.h, self._slot)
        if have < 0:
            _capi.check(have, "edmp_guide_slot")
        if have == 1:
            ctx.bound_guide = self
            return
        ctx.bound_guide = None
        no = self.obstacle_config.shape[0]
        _capi.check(
            ctx.lib.edmp_scene_set(ctx.h, _capi.as_pd(self.obstacle_config), no, _capi.as_pd(self._cls_clr), _capi.as_pd(self._cls_exp),
                                   self._cls_clr.shape[0], self.T, _capi.as_pf(self._half), _capi.as_pf(self._dh), _capi.as_pf(self._sf)),
            "edmp_scene_set",
        )
        self._rows_token = None
        self._set_rows(self._sched)
        if self._kinds is not None:
            _capi.check(ctx.lib.edmp_scene_set_shapes(ctx.h, _capi.as_pi32(self._kinds), no), "edmp_scene_set_shapes")
        ctx.bound_guide = self  # only a completely built object counts as bound

    def _check_kinds(self, kinds):
        k = np.ascontiguousarray(np.asarray(kinds, dtype=np.int32).reshape(-1))
        if k.shape[0] != self.obstacle_config.shape[0] or not np.all((k == 0) | (k == 1)):
            raise ValueError("obstacle_kinds: one entry per obstacle, 0 = cuboid, 1 = cylinder")
        return k

    def set_obstacle_kinds(self, kinds):
        """mark obstacle rows as true cylinders (dims = (r, r, h)) for the success check (lib/environment.py:249-268)."""
        self._kinds = self._check_kinds(kinds)
        self._bind()
        _capi.check(self.ctx.lib.edmp_scene_set_shapes(self.ctx.h, _capi.as_pi32(self._kinds), self._kinds.shape[0]), "edmp_scene_set_shapes")

    def _set_rows(self, sched):
        sched = np.ascontiguousarray(np.asarray(sched, dtype=np.float64))
        token = (sched.shape, sched.tobytes())
        if self._rows_token == token:
            return
        ctx = self.ctx
        method = np.ascontiguousarray(np.asarray(self.guide_cfgs["guidance_method"], dtype=np.float32))
        gn = np.ascontiguousarray(np.asarray(self.guide_cfgs["grad_norm"], dtype=np.float64))
        _capi.check(
            ctx.lib.edmp_rows_set(ctx.h, _capi.as_pi32(self.row_class), _capi.as_pf(method), _capi.as_pd(gn), _capi.as_pd(sched), self.batch_size, sched.shape[1]),
            "edmp_rows_set",
        )
        self._rows_token = token

    # ---- reference API -----------------------------------------------------------------------------------------
    def define_obstacles(self, obstacle_config=None, t=0, batch_size=None):
        """sets self.obs_min / self.obs_max (b, no, 3) like lib/guide.py:118-158 (read back from the device table)."""
        self._bind()
        b = self.batch_size if batch_size is None else int(batch_size)
        if t != 0 and b != self.batch_size:
            raise ValueError("t != 0 needs batch_size == total batch (the reference broadcasts per-row schedules)")
        no = self.obstacle_config.shape[0]
        tabs = {}
        out = np.zeros((b, no, 6), dtype=np.float32)
        for r in range(b):
            cls = int(self.row_class[r]) if t != 0 else 0
            if cls not in tabs:
                buf = np.zeros((no, 6), dtype=np.float32)
                _capi.check(self.ctx.lib.edmp_scene_read_aabbs(self.ctx.h, cls, int(t), _capi.as_pf(buf)))
                tabs[cls] = buf
            out[r] = tabs[cls]
        self.obs_min = torch.from_numpy(out[:, :, :3].copy())
        self.obs_max = torch.from_numpy(out[:, :, 3:].copy())

    def _cost_common(self, joint_tensor, t, batch_size):
        self._bind()
        jt = self.ctx.to_dev(joint_tensor, torch.float32)
        if jt.dim() != 3 or jt.shape[1] != 7:
            raise ValueError(f"joint tensor must be (n, 7, L), got {tuple(jt.shape)}")
        n, L = jt.shape[0], jt.shape[2]
        b = self.batch_size if batch_size is None else int(batch_size)
        if b != n:
            raise ValueError(f"batch_size ({b}) must equal the number of joint rows ({n})")
        use_rows = 0
        if t != 0:
            if n != self.batch_size:
                raise ValueError("t != 0 needs n == total batch (per-row inflation schedules)")
            use_rows = 1
        return jt, n, L, use_rows

    def cost(self, joint_tensor, t, batch_size=None):
        jt, n, L, use_rows = self._cost_common(joint_tensor, t, batch_size)
        no = self.obstacle_config.shape[0]
        vol = self.ctx.empty((n, L, 9 * no), torch.float32)
        _capi.check(self.ctx.lib.edmp_guide_cost_dev(self.ctx.h, ptr(jt), n, L, int(t), use_rows, ptr(vol)), "edmp_guide_cost_dev")
        self.ctx.sync()
        return vol

    def swept_volume_cost(self, joint_tensor, start, goal, t, batch_size=None):
        jt, n, L, use_rows = self._cost_common(joint_tensor, t, batch_size)
        no = self.obstacle_config.shape[0]
        s = np.ascontiguousarray(np.asarray(start.detach().cpu() if isinstance(start, torch.Tensor) else start, dtype=np.float32).reshape(7))
        g = np.ascontiguousarray(np.asarray(goal.detach().cpu() if isinstance(goal, torch.Tensor) else goal, dtype=np.float32).reshape(7))
        vol = self.ctx.empty((n, L + 1, 9 * no), torch.float32)
        _capi.check(self.ctx.lib.edmp_guide_swept_cost_dev(self.ctx.h, ptr(jt), n, L, int(t), use_rows, _capi.as_pf(s), _capi.as_pf(g), ptr(vol)),
                    "edmp_guide_swept_cost_dev")
        self.ctx.sync()
        return vol

    def get_gradient(self, joint_input, start, goal, t):
        self._bind()
        ctx = self.ctx
        ji = ctx.to_dev(np.asarray(joint_input, dtype=np.float64), torch.float64)
        B, L = ji.shape[0], ji.shape[2]
        s = np.ascontiguousarray(np.asarray(start, dtype=np.float64).reshape(7))
        g = np.ascontiguousarray(np.asarray(goal, dtype=np.float64).reshape(7))
        out = ctx.empty((B, 7, L), torch.float64)
        _capi.check(ctx.lib.edmp_guide_gradient_dev(ctx.h, ptr(ji), B, L, _capi.as_pd(s), _capi.as_pd(g), int(t), ptr(out), None), "edmp_guide_gradient_dev")
        return ctx.to_host(out)

    def row_swept_volumes(self, start, goal, trajectories):
        """(B,) f32 t=0 swept volume per row and the argmin (first on ties)."""
        self._bind()
        ctx = self.ctx
        if isinstance(trajectories, torch.Tensor) and trajectories.is_cuda:
            X = ctx.adopt(trajectories.to(torch.float64).contiguous())
        else:
            X = ctx.to_dev(np.asarray(trajectories, dtype=np.float64), torch.float64)
        B, N = X.shape[0], X.shape[2]
        s = np.ascontiguousarray(np.asarray(start, dtype=np.float64).reshape(7))
        g = np.ascontiguousarray(np.asarray(goal, dtype=np.float64).reshape(7))
        vols = ctx.empty((B,), torch.float32)
        idx = C.c_int()
        _capi.check(ctx.lib.edmp_row_swept_volumes_dev(ctx.h, ptr(X), B, N, _capi.as_pd(s), _capi.as_pd(g), ptr(vols), C.byref(idx)), "edmp_row_swept_volumes_dev")
        return ctx.to_host(vols), idx.value

    def success_rows(self, trajectories, substeps: int = 4, return_device: bool = False):
        """Geometric success of EVERY row (edmp_success_rows_dev; stands for RobotEnvironment.benchmark_trajectory,
        lib/environment.py:632-680, and the tally of infer_serial.py:94-99,165-168): trajectories (B,7,N) ndarray or device
        tensor -> dict(ok (B,) bool, first (B,) int32 first colliding waypoint or -1, within (B,) bool, collision_free (B,) bool,
        rows_ok, rows_within, rows_collision_free, rows).  ``collision_free`` (= first < 0) is the REFERENCE's success flag: its
        benchmark_trajectory fails a plan on contact only and merely prints when the joint limits are left
        (lib/environment.py:659-661, 672); ``ok`` = collision_free AND within, the stricter flag.  With return_device the per-row
        arrays stay device tensors (ok / first / within int32, collision_free bool)."""
        self._bind()
        ctx = self.ctx
        if isinstance(trajectories, torch.Tensor) and trajectories.is_cuda:
            X = ctx.adopt(trajectories.to(torch.float64).contiguous())
        else:
            X = ctx.to_dev(np.asarray(trajectories, dtype=np.float64), torch.float64)
        if X.dim() != 3 or X.shape[1] != 7:
            raise ValueError(f"trajectories must be (B, 7, N), got {tuple(X.shape)}")
        B, N = X.shape[0], X.shape[2]
        flags = ctx.empty((3, B), torch.int32)
        counts = (C.c_int32 * 4)()
        dh = np.ascontiguousarray(franka.dh_table_f64())
        _capi.check(ctx.lib.edmp_success_rows_dev(ctx.h, ptr(X), B, N, int(substeps), _capi.as_pd(dh), C.c_void_p(flags[0].data_ptr()),
                                                  C.c_void_p(flags[1].data_ptr()), C.c_void_p(flags[2].data_ptr()), counts), "edmp_success_rows_dev")
        out = dict(rows_ok=int(counts[0]), rows_within=int(counts[1]), rows_collision_free=int(counts[2]), rows=int(counts[3]))
        if return_device:
            with torch.cuda.stream(ctx.stream):
                cf = flags[1] < 0
            out.update(ok=flags[0], first=flags[1], within=flags[2], collision_free=cf)
        else:
            f = ctx.to_host(flags)
            out.update(ok=f[0].astype(bool), first=f[1].copy(), within=f[2].astype(bool), collision_free=f[1] < 0)
        return out

    def choose_best_trajectory(self, start, goal, trajectories):
        _, idx = self.row_swept_volumes(start, goal, trajectories)
        return trajectories[idx]
