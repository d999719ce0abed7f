"""Multi-GPU data path: one process per GPU, batch rows sharded across ranks, no collective inside the reverse loop.

The reference has no distributed code: its only coupling between rows is the whole-batch gradient
norm (lib/guide.py:629).  Two modes:
  * "replicas" (default, used by bench.py): every rank runs its OWN reference batch (own rows, own noise) — exactly
    what launching the reference once per GPU does — and the ranks only meet at the end of sampling to agree on
    the best trajectory: an all-gather of (swept volume, local argmin, success flag) and a broadcast of the winning
    (7, 50) trajectory from its owner.  Messages are a few hundred bytes: latency-bound on xGMI.
  * "one logical batch" (parity with a single-process run of B_total rows): additionally all-reduce one f64
    (sum g^2) per guided step between step_a and step_b (Diffusion.denoise_step(allreduce=...)).
Backends: "nccl" (= RCCL on ROCm) with device tensors, "gloo" with host tensors (CPU tests).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def shard_rows(total_rows: int, rank: int, world: int):
    """rows [lo, hi) owned by `rank`: floor split, contiguous."""
    lo = (rank * total_rows) // world
    hi = ((rank + 1) * total_rows) // world
    return lo, hi


def shard_guide_cfgs(cfgs: dict, lo: int, hi: int) -> dict:
    out = dict(cfgs)
    for k in ("clearance", "expansion", "guidance_method", "grad_norm", "guidance_schedule", "volume_trust_region"):
        out[k] = np.ascontiguousarray(cfgs[k][lo:hi])
    out["total_batch_size"] = hi - lo
    return out


def _comm_device(device=None):
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_best(local_volume: float, local_index: int, local_traj, success: bool, device=None, group=None, always=False, rows_ok: int = 0, rows: int = 0,
                collision_free=None, rows_collision_free: int = 0):
    """End-of-sampling exchange.  Returns dict(volume, rank, index, traj (7,50) f64 ndarray, success, n_success, rows_ok,
    rows).  ``success`` is the flag of the winning row; ``rows_ok`` / ``rows`` are this rank's batch success counts
    (IntersectionVolumeGuide.success_rows) and come back SUMMED over the ranks - the job-wide plan success rate.
    ``collision_free`` / ``rows_collision_free``: the same pair under the REFERENCE's criterion - no contact, joint limits only
    printed (lib/environment.py:659-661, 672) - where ``success`` / ``rows_ok`` also require every waypoint inside the limits;
    ``collision_free=None`` (callers that predate the split) reports the strict flag for both.
    Ties resolve to the lowest rank (= lowest global row index, like torch.argmin over the unsharded batch).
    ``always``: run the collectives even in a world of one (exercises the RCCL path on a single-GPU box)."""
    traj = np.asarray(local_traj, dtype=np.float64)
    cfree = bool(success) if collision_free is None else bool(collision_free)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not always):
        return dict(volume=float(local_volume), rank=0, index=int(local_index), traj=traj.copy(), success=bool(success), n_success=int(bool(success)),
                    rows_ok=int(rows_ok), rows=int(rows), collision_free=cfree, rows_collision_free=int(rows_collision_free))
    dev = _comm_device(device)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = torch.tensor([float(local_volume), float(local_index), 1.0 if success else 0.0, float(rows_ok), float(rows), 1.0 if cfree else 0.0, float(rows_collision_free)],
                        dtype=torch.float64, device=dev)
    allv = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine, group=group)
    table = torch.stack(allv).cpu().numpy()
    vols = table[:, 0]
    owner = int(np.argmin(vols))  # first minimum -> lowest rank; NaN wins like torch.argmin over the unsharded batch (np.argmin agrees)
    t = torch.from_numpy(traj.copy()).to(dev) if rank == owner else torch.empty(traj.shape, dtype=torch.float64, device=dev)
    dist.broadcast(t, src=owner if group is None else dist.get_global_rank(group, owner), group=group)
    return dict(volume=float(vols[owner]), rank=owner, index=int(table[owner, 1]), traj=t.cpu().numpy(), success=bool(table[owner, 2] > 0),
                n_success=int((table[:, 2] > 0).sum()), rows_ok=int(table[:, 3].sum()), rows=int(table[:, 4].sum()), collision_free=bool(table[owner, 5] > 0),
                rows_collision_free=int(table[:, 6].sum()))


def allreduce_sum_(t: torch.Tensor, group=None, always=False):
    """in-place sum over ranks (the per-guided-step scalar of the 'one logical batch' mode).  With the nccl backend the
    collective is enqueued by RCCL behind the CURRENT torch stream's work (Diffusion.denoise_guided makes the context's
    stream current around the call).  ``always``: issue the collective even in a world of one."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or always):
        if dist.get_backend() == "nccl":
            dist.all_reduce(t, group=group)
        else:
            c = t.detach().cpu()
            dist.all_reduce(c, group=group)
            t.copy_(c.to(t.device))
    return t


def _rccl_library_path():
    """the librccl the process already mapped (PyTorch-ROCm ships one beside libtorch_hip.so), from /proc/self/maps; None if none"""
    try:
        for line in open("/proc/self/maps"):
            path = line.rsplit(None, 1)[-1]
            if "/librccl.so" in path:
                return path
    except OSError:
        pass
    return None


class RcclAllReduce:
    """The per-guided-step all-reduce as NATIVE code (csrc/rccl_hook.hip): ``dif.denoise_guided(allreduce=RcclAllReduce(dif))``
    lets the device-resident loop call ncclAllReduce itself - no Python callback, no GIL, on the loop's path (the callable form
    ``allreduce=dist.allreduce_sum_`` stays: gloo test boxes, other collectives).

    One communicator per context, created ONCE: rank 0 draws the ncclUniqueId, the 128 bytes travel through the default
    torch.distributed group (any backend), every rank calls ncclCommInitRank on its context's device.  ``comm_ptr``: borrow an
    existing communicator instead (torch: ``pg._get_backend(torch.device("cuda"))._comm_ptr()``).  In a world of one (or with
    no process group) the communicator has one rank: the collective is still issued, which is what N = 1 measures."""

    def __init__(self, diffusion, group=None, comm_ptr=None, library=None):
        import ctypes as C

        from . import _capi

        ctx = diffusion.ctx  # (a Diffusion: the sampler state the hook lives in is created from its T / variance threshold)
        ctx.ensure_sampler(diffusion.T, diffusion.variance_thresh)
        self.ctx, self.lib = ctx, ctx.lib
        _capi.check(self.lib.edmp_rccl_load((library or _rccl_library_path() or "").encode() or None), "edmp_rccl_load")
        if comm_ptr is not None:
            _capi.check(self.lib.edmp_rccl_attach_comm(ctx.h, C.c_void_p(int(comm_ptr))), "edmp_rccl_attach_comm")
        else:
            on = dist.is_available() and dist.is_initialized()
            world = dist.get_world_size(group) if on else 1
            rank = dist.get_rank(group) if on else 0
            buf = C.create_string_buffer(128)
            if rank == 0:
                _capi.check(self.lib.edmp_rccl_unique_id(buf), "edmp_rccl_unique_id")
            box = [bytes(buf.raw)]
            if world > 1:
                dist.broadcast_object_list(box, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
            ident = C.create_string_buffer(box[0], 128)
            with torch.cuda.device(ctx.device):
                _capi.check(self.lib.edmp_rccl_attach(ctx.h, ident, world, rank), "edmp_rccl_attach")
        info = (C.c_int32 * 3)()
        _capi.check(self.lib.edmp_rccl_info(ctx.h, info))
        self.world, self.rank, self.kind = int(info[0]), int(info[1]), {1: "own communicator", 2: "borrowed communicator"}[int(info[2])]
        _capi.check(self.lib.edmp_rccl_enable(ctx.h, 0))  # denoise_guided(allreduce=self) switches it on for its own run only

    def attached(self) -> bool:
        import ctypes as C

        info = (C.c_int32 * 3)()
        self.lib.edmp_rccl_info(self.ctx.h, info)
        return int(info[2]) != 0

    def close(self):
        from . import _capi

        if self.ctx.h:
            _capi.check(self.lib.edmp_rccl_detach(self.ctx.h), "edmp_rccl_detach")


def geometric_success(volume: float, traj, lo=None, hi=None) -> bool:
    """success PROXY (pybullet is not available): zero t=0 swept volume and all waypoints inside the joint limits
    (SURVEY.md §8d).  Not the paper's success rate."""
    from .franka import joint_limits

    if lo is None:
        lo, hi = joint_limits()
    tr = np.asarray(traj)
    return bool(volume == 0.0 and np.all(tr >= lo[:, None] - 1e-9) and np.all(tr <= hi[:, None] + 1e-9))
