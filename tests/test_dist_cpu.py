"""CPU, world_size 2, gloo: the multi-GPU data path's host logic (row sharding, end-of-sampling gather, the optional
per-step scalar all-reduce) — the same functions bench.py drives over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import cfgs_for


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from edmp_amd import dist as ED

    total = 13
    lo, hi = ED.shard_rows(total, rank, world)
    vols = np.array([5.0, 3.0, 9.0, 3.0, 7.0, 2.5, 8.0, 2.5, 6.0, 4.0, 1.5, 1.5, 9.0])  # global min 1.5 first at row 10
    trajs = np.arange(total)[:, None, None] * np.ones((total, 7, 50))
    li = int(np.argmin(vols[lo:hi]))
    res = ED.gather_best(float(vols[lo + li]), li, trajs[lo + li], success=(rank == 1), rows_ok=3 + rank, rows=hi - lo, collision_free=True, rows_collision_free=4 + rank)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    ED.allreduce_sum_(t)
    cfgs = cfgs_for([1, 10, 11], 4)
    sh = ED.shard_guide_cfgs(cfgs, *ED.shard_rows(12, rank, world))
    q.put((rank, lo, hi, res["rank"], res["index"], res["volume"], float(res["traj"][0, 0]), res["n_success"], float(t.item()),
           sh["total_batch_size"], sh["guidance_method"].tolist(), res["rows_ok"], res["rows"], res["success"], res["rows_collision_free"], res["collision_free"]))
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, *rest0), (r1, lo1, hi1, *rest1) = out
    assert (lo0, hi0, lo1, hi1) == (0, 6, 6, 13)
    # both ranks agree: winner = rank 1, local row 4 (global row 10), volume 1.5, that row's trajectory
    assert rest0[:5] == rest1[:5] == [1, 4, 1.5, 10.0, 1]
    assert rest0[5] == rest1[5] == 3.0  # all-reduced scalar
    assert rest0[6] == 6 and rest1[6] == 6
    assert rest0[7] == [0, 0, 0, 0, 1, 1] and rest1[7] == [1, 1, 1, 1, 1, 1]
    # batch success counts come back summed over the ranks (3 + 4 of 6 + 7 rows); `success` is the winning row's flag (rank 1's)
    # the reference's criterion (collision-free, limits not required) travels beside the strict one: 4 + 5 rows, winning row's flag
    assert rest0[8:] == rest1[8:] == [7, 13, True, 9, True]


def test_gather_single_process_passthrough():
    from edmp_amd import dist as ED

    r = ED.gather_best(0.25, 3, np.ones((7, 50)), True)
    assert r["rank"] == 0 and r["index"] == 3 and r["volume"] == 0.25 and r["n_success"] == 1 and (r["rows_ok"], r["rows"]) == (0, 0)
    r = ED.gather_best(0.25, 3, np.ones((7, 50)), False, rows_ok=17, rows=1024)
    assert (r["rows_ok"], r["rows"], r["success"]) == (17, 1024, False) and r["collision_free"] is False  # callers without the split: strict flag for both
    r = ED.gather_best(0.25, 3, np.ones((7, 50)), False, rows_ok=17, rows=1024, collision_free=True, rows_collision_free=40)
    assert (r["success"], r["collision_free"], r["rows_collision_free"]) == (False, True, 40)
    assert ED.shard_rows(1024, 3, 8) == (384, 512)


def test_geometric_success_proxy():
    from edmp_amd import dist as ED
    from edmp_amd.franka import joint_limits

    lo, hi = joint_limits()
    inside = np.tile(((lo + hi) / 2)[:, None], (1, 50))
    assert ED.geometric_success(0.0, inside)
    assert not ED.geometric_success(1e-6, inside)
    outside = inside.copy()
    outside[3, 10] = 0.5  # joint 4 upper limit is -4 deg
    assert not ED.geometric_success(0.0, outside)


def _summary_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import infer_serial

    # rank r planned the scenes i with i % world == r of a five-scene problem set (infer_serial.run's deal): 3 + 2
    mine = [dict(success_proxy=int(i % 3 == 0), success_strict=0, rows_collision_free=i, rows=4, planning_time_s=0.25) for i in range(5) if i % world == rank]
    q.put((rank, infer_serial.job_summary(mine, world)))
    dist.destroy_process_group()


def test_scene_sharded_driver_summary_world2():
    """infer_serial under a launcher: scenes are dealt round-robin, only the tallies meet at the end (all_gather_object)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_summary_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = dict(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = dict(scenes=5, success_proxy=2, success_strict=0, rows_collision_free=10, rows=20, planning_time_s=1.25, ranks=2, scenes_per_rank=[3, 2])
    assert out[0] == out[1] == want
