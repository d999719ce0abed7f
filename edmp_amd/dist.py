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


def geometric_success(volume: float, traj, lo=None, hi=None) -> bool:
    """success PROXY (pybullet is not available): zero t=0 swept volume and all waypoints inside the joint limits
    (SURVEY.md §8d).  Not the paper's success rate."""
    from .franka import joint_limits

    if lo is None:
        lo, hi = joint_limits()
    tr = np.asarray(traj)
    return bool(volume == 0.0 and np.all(tr >= lo[:, None] - 1e-9) and np.all(tr <= hi[:, None] + 1e-9))
