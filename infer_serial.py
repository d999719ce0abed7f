#!/usr/bin/env python
"""infer_serial.py — the reference's driver surface (infer_serial.py:14-170) on the MI355X-native sampler.

    python infer_serial.py -c configs/cfg_c1_plumbing.yaml

Same flow: run config (YAML, schema of benchmark/cfgs/cfg1.yaml) -> guide plugins (guides/cfgs/guide<N>.yaml schema)
-> per-row guide_cfgs -> per scene: guide object, IK-goal filter (guide.cost at t=0, trust region 0.0008, nearest to
start), Diffusion.denoise_guided, choose_best_trajectory.  Differences forced by missing third-party assets, all
stated in DESIGN.md: scenes/IK goals are synthetic unless a dataset object is supplied, weights are random-init
unless <model_dir>/TemporalUNetModel<T>_N<traj_len>/weights_latest.pt exists, success is the geometric proxy
(pybullet absent)."""
import argparse
import os
import time

import numpy as np
import torch

from edmp_amd import dist as ED
from edmp_amd import evaluation as EV
from edmp_amd import guide_cfg as GC
from edmp_amd.diffusion import Diffusion
from edmp_amd.guide import IntersectionVolumeGuide
from edmp_amd.scenes import SyntheticDataset
from edmp_amd.temporalunet import TemporalUNet


def _ranks(device):
    """Launched under ``python -m torch.distributed.run --nproc-per-node N infer_serial.py ...`` (one process per GPU): join the
    process group and return (rank, world, this rank's device).  Backend "nccl" (= RCCL) when every local rank has its own GPU,
    EDMP_DIST_BACKEND=gloo lets the ranks share one (single-GPU test boxes), as in bench.py.  Outside a launcher: (0, 1, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, device
    import torch.distributed as dist

    backend = os.environ.get("EDMP_DIST_BACKEND", "nccl")
    local, ngpu = int(os.environ.get("LOCAL_RANK", "0")), torch.cuda.device_count()
    if backend == "nccl" and ngpu < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
        raise SystemExit(f"[infer_serial] {world} ranks over RCCL need one GPU each, this node shows {ngpu}: launch fewer ranks (or EDMP_DIST_BACKEND=gloo to share)")
    index = local if backend == "nccl" else local % max(ngpu, 1)
    torch.cuda.set_device(index)
    if not dist.is_initialized():
        dist.init_process_group(backend, **({"device_id": torch.device("cuda", index)} if backend == "nccl" else {}))
    return dist.get_rank(), dist.get_world_size(), f"cuda:{index}"


class _NoiseFeeder:
    """Every scene's (T+1, B, C, N) noise stream, drawn IN SCENE ORDER from the global NumPy RandomState (so each
    scene sees the numbers the serial loop would give it) by ONE background thread into page-locked whole-scene buffers, ahead of the
    scene that consumes it; the planning threads upload their buffer by DMA.  Drawing on the calling thread instead serialised the
    loop on the host: 0.16 s of draws + a pageable 734 MB upload per 1024-row scene made two scenes in flight SLOWER than the serial
    loop (profiles/r05_problem_set.md).  Exactly `n_scenes` streams are drawn: the global state ends where the serial loop leaves it."""

    def __init__(self, ctx, n_scenes, shape, n_buffers):
        import queue
        import threading

        from edmp_amd import nprng

        n = int(np.prod(shape))
        cache = getattr(ctx, "_scene_noise_buffers", None)
        if cache is None or len(cache) < n_buffers or cache[0].numel() != n:
            cache = ctx._scene_noise_buffers = [torch.empty(n, dtype=torch.float64, pin_memory=True) for _ in range(n_buffers)]
        self.shape, self.free, self.ready = tuple(shape), queue.Queue(), queue.Queue()
        for b in cache[:n_buffers]:
            self.free.put(b)
        nthr = nprng.draw_threads()

        from edmp_amd.diffusion import PinnedNoiseStream

        per_step = int(np.prod(shape[1:]))
        pieces, left, kk, first = [], int(shape[0]) - 1, 1, True  # (X_T + 1 step), 2, 4, 8, 16, 16, ... steps: the sampler's own chunk plan
        while left > 0:
            st = min(kk, 16, left)
            pieces.append((st + (1 if first else 0)) * per_step)
            left, kk, first = left - st, kk * 2, False

        def work():
            stream = None
            try:
                for _ in range(n_scenes):
                    b = self.free.get()
                    if b is None:
                        return
                    stream = PinnedNoiseStream(b.view(self.shape))
                    self.ready.put(stream)  # handed out at once: the consumer uploads each chunk as soon as it is drawn
                    flat, off = b.numpy(), 0
                    for m in pieces:  # piecewise draws continue the legacy stream exactly (the cached second gauss value travels in the state)
                        nprng.standard_normal((m,), nthreads=nthr, out=flat[off:off + m])
                        off += m
                        stream.publish(off)
            except BaseException as exc:  # surfaced by next() / by the consumer's wait
                if stream is not None:
                    stream.publish(stream.drawn, error=exc)
                self.ready.put(exc)

        self.thread = threading.Thread(target=work, name="edmp-scene-noise", daemon=True)
        self.thread.start()

    def next(self):
        """the next scene's stream (scene order): a PinnedNoiseStream over a pinned (T+1, B, C, N) f64 tensor, possibly still being drawn"""
        b = self.ready.get()
        if isinstance(b, BaseException):
            raise b
        return b

    def recycle(self, stream):
        self.free.put(stream.tensor.view(-1))

    def close(self):
        self.free.put(None)
        self.thread.join()


def job_summary(results, world=1):
    """This rank's tallies; under a launcher the sum over all ranks (all_gather_object of five small integers per rank - the
    trajectories stay where they were planned).  Keys: scenes, success_proxy (the reference's tally), success_strict, rows_collision_free, rows."""
    mine = dict(scenes=len(results), success_proxy=sum(r["success_proxy"] for r in results), success_strict=sum(r["success_strict"] for r in results),
                rows_collision_free=sum(r["rows_collision_free"] for r in results), rows=sum(r["rows"] for r in results),
                planning_time_s=float(sum(r["planning_time_s"] for r in results)))
    if world <= 1:
        return dict(mine, ranks=1)
    import torch.distributed as dist

    parts = [None] * world
    dist.all_gather_object(parts, mine)
    out = {k: sum(p[k] for p in parts) for k in mine}
    out["ranks"] = world
    out["scenes_per_rank"] = [p["scenes"] for p in parts]
    return out


def run(cfg_path, dataset=None, max_scenes=None, verbose=True, scenes_in_flight=1, shard_scenes=True):
    """The reference's scene loop (infer_serial.py:95-170).  Under ``torch.distributed.run`` (one process per GPU, extension: the
    reference is one process) the scenes are dealt round-robin to the ranks - scene i of the cfg's order goes to rank i mod world -
    and nothing is exchanged until `job_summary` adds the tallies up: scenes are independent problems, this is the problem set's natural
    shard (SURVEY.md 8e "replicas + final gather").  Every rank draws from its OWN process-global NumPy RandomState, as N separate
    runs of the reference would (the reference never seeds it, infer_serial.py has no np.random.seed).  ``scenes_in_flight`` > 1 (an extension; the reference is serial) plans
    that many scenes concurrently on one GPU, each on its own context / stream / host thread: every launch of the sampler is one
    wave of 256 workgroups, a second independent scene fills its dispatch gaps and kernel tails (+7-9 % throughput measured,
    bench.py: two_scenes_in_flight).  Per-scene results are identical to the serial loop's: scenes are prepared in order on the
    calling thread and each scene's noise is drawn - by ONE background feeder thread, a whole scene ahead, serial loop included - from the
    global NumPy RandomState in scene order.  While a run is in progress nothing else may draw from (or seed) the global RandomState: the
    feeder reads and advances it from its own thread (np.random.get_state / set_state are not atomic)."""
    from concurrent.futures import ThreadPoolExecutor

    from edmp_amd.runtime import get_context, lane_context

    t_enter = time.time()
    benchmark_cfg = GC.load_yaml(cfg_path)
    rank, world, device = _ranks(benchmark_cfg["model"]["device"]) if shard_scenes else (0, 1, benchmark_cfg["model"]["device"])
    traj_len = benchmark_cfg["model"]["traj_len"]
    T = benchmark_cfg["model"]["T"]
    num_channels = benchmark_cfg["model"]["num_channels"]
    if dataset is None:
        # the reference's own dataset types (datasets/load_test_dataset.py:15-38: 'global' | 'hybrid' | 'both' -> <path>/<type>_solvable_problems.pkl):
        # served from the JSON scripts/mpinets_pkl_to_json.py writes next to the pickle, when it is there
        dtype, dpath = benchmark_cfg["dataset"]["dataset_type"], benchmark_cfg["dataset"]["path"]
        converted = os.path.join(dpath, f"{dtype}_solvable_problems.json")
        if dtype in ("global", "hybrid", "both"):
            if not os.path.exists(converted):
                raise FileNotFoundError(f"{converted} not found: convert {dtype}_solvable_problems.pkl with scripts/mpinets_pkl_to_json.py (IK goals via --ik-goals), "
                                        "or use dataset_type: 'synthetic'")
            from edmp_amd.scenes import ProblemSetDataset

            dataset = ProblemSetDataset(converted)
    if dataset is None:
        dataset = SyntheticDataset(benchmark_cfg["dataset"]["dataset_type"], d_path=benchmark_cfg["dataset"]["path"],
                                   scene_types=benchmark_cfg["dataset"]["scene_types"],
                                   num_scenes_per_type=benchmark_cfg["dataset"].get("num_scenes_per_type", 1))
    guide_cfgs = GC.guide_cfgs_from_run_cfg(benchmark_cfg, base_dir=os.path.dirname(os.path.abspath(cfg_path)) + "/..")
    total_batch_size = guide_cfgs["total_batch_size"]
    model_name = benchmark_cfg["model"]["model_dir"] + "TemporalUNetModel" + str(T) + "_N" + str(traj_len)
    if not os.path.exists(model_name):
        if verbose and rank == 0:
            print(f"[infer_serial] {model_name} not found: using a seeded random-init denoiser (no trained weights offline)")
        model_name = None
    k = max(1, int(scenes_in_flight))
    base = get_context(device)
    lanes = []  # one (context, diffuser, denoiser) per scene in flight; the weights are replicated per context
    for j in range(k):
        ctx = lane_context(base, j)  # lane 0 = the device's context; further lanes are cached per (device, lane), not re-created per call
        lanes.append((Diffusion(T=T, device=ctx), TemporalUNet(model_name=model_name, input_dim=num_channels, time_dim=32, dims=(32, 64, 128, 256, 512, 512),
                                                                device=ctx, max_batch=total_batch_size)))

    def plan(lane, guide, start_joints, goal_joints, noise, meta, t0):
        diffuser, denoiser = lanes[lane]
        tm = dict(meta.pop("timings"))
        pinned = None
        if noise is not None:  # this scene's stream, drawn ahead by the feeder into page-locked memory: uploaded in chunks beside the loop
            t_w = time.time()
            pinned = noise.result() if hasattr(noise, "result") else noise
            tm["noise_wait_s"] = time.time() - t_w
            noise = pinned
        ta = time.time()
        trajectories = diffuser.denoise_guided(model=denoiser, guide=guide, batch_size=total_batch_size, traj_len=traj_len,
                                               num_channels=num_channels, condition=True, benchmarking=True, start=start_joints,
                                               goal=goal_joints, guidance_schedule=guide_cfgs["guidance_schedule"], noise=noise)
        tb = time.time()
        if pinned is not None:
            feeder.recycle(pinned)  # (denoise_guided returned host trajectories: the upload out of the buffer is long done)
        vols, idx = guide.row_swept_volumes(start_joints, goal_joints, trajectories)
        trajectory = trajectories[idx]
        t_plan = time.time() - t0
        tm["denoise_s"], tm["best_trajectory_s"] = tb - ta, time.time() - tb
        # success: pybullet execution (lib/environment.py:632-680) is unavailable -> exact link-box vs cuboid / cylinder
        # check along the interpolated trajectory, for EVERY row of the batch in one kernel (csrc/success.hip); the
        # scene's success is the chosen row's flag (infer_serial.py:165-168) under the reference's rule - no contact;
        # leaving the joint limits only prints there (lib/environment.py:659-661, 672) -, the stricter flag (also inside the
        # limits) and the batch rates are reported next to it, as is the guide's own (conservative, AABB) criterion
        tc = time.time()
        chk = guide.success_rows(trajectories)
        tm["success_check_s"] = time.time() - tc
        return dict(**meta, timings=tm, best_row=int(idx), swept_volume=float(vols[idx]), success_proxy=int(chk["collision_free"][idx]), success_strict=int(chk["ok"][idx]),
                    rows_collision_free=chk["rows_collision_free"], rows_ok=chk["rows_ok"], rows=chk["rows"],
                    aabb_volume_zero=bool(ED.geometric_success(float(vols[idx]), trajectory)), first_collision_waypoint=int(chk["first"][idx]),
                    path_length=EV.path_lengths(trajectory), sparc=EV.smoothness(trajectory), planning_time_s=t_plan, scene_wall_s=time.time() - t0, trajectory=trajectory)

    t_success, t_strict, i, results, pending = 0, 0, 0, [], []

    def collect(fut):
        nonlocal t_success, t_strict
        r = fut.result() if hasattr(fut, "result") else fut
        r["done_at"] = time.time()
        results.append(r)
        t_success += r["success_proxy"]  # the reference's tally: collision-free (infer_serial.py:165-168 on lib/environment.py:672)
        t_strict += r["success_strict"]
        if verbose:
            print(("" if world == 1 else f"[rank {rank}] ") + f"Scene {len(results)} ({r['scene_type']}/{r['scene_num']}): planning {r['planning_time_s']:.2f} s, best row {r['best_row']}, swept volume "
                  f"{r['swept_volume']:.4g}, geometric success (proxy, collision-free) {r['success_proxy']} ({r['rows_collision_free']}/{r['rows']} rows of the batch); "
                  f"also within the joint limits {r['success_strict']} ({r['rows_ok']}/{r['rows']})   running {t_success}/{len(results)} (strict {t_strict}/{len(results)})")

    # this rank's scenes, in the cfg's order (scene i of that order belongs to rank i mod world)
    mine = []
    for scene_type in benchmark_cfg["dataset"]["scene_types"]:
        for scene_num in range(dataset.data_nums[scene_type]):
            if max_scenes is not None and i >= max_scenes:
                break
            if i % world == rank:
                mine.append((i, scene_type, scene_num))
            i += 1
    # the noise of EVERY scene comes from the feeder thread, one whole scene ahead (round 6: the serial loop too - drawing chunk by chunk
    # beside the GPU had no margin left once a reverse step took 0.92 ms: a slower host capped the scene loop, BENCH_r05 0.947 x value)
    feeder = _NoiseFeeder(base, len(mine), (T + 1, total_batch_size, num_channels, traj_len), k + 1) if mine else None
    run.last_setup_s = time.time() - t_enter  # config, dataset, model load / upload: per run, not per scene
    try:
        with ThreadPoolExecutor(max_workers=k) as pool:
            for i, scene_type, scene_num in mine:
                lane = (i // world) % k
                while len(pending) >= k:  # the lane's previous scene (and every earlier one) is done before its context is reused
                    collect(pending.pop(0))
                obstacle_config, _, _, num_cuboids, num_cylinders, start_joints, all_ik_goals = dataset.fetch_data(scene_num=scene_num, scene_type=scene_type)
                t0 = time.time()
                # obstacle_config = cuboids first, then cylinders as (r, r, h) boxes (datasets/load_test_dataset.py:141-149); the
                # success check spawns the latter as true cylinders (infer_serial.py:159-163 -> lib/environment.py:249-268)
                kinds = np.concatenate([np.zeros(int(num_cuboids), dtype=np.int32), np.ones(int(num_cylinders), dtype=np.int32)])
                guide = IntersectionVolumeGuide(obstacle_config=obstacle_config, device=lanes[lane][0].ctx, guide_cfgs=guide_cfgs, batch_size=total_batch_size,
                                                obstacle_kinds=kinds, mesh_dir=benchmark_cfg["model"].get("mesh_dir"))
                t1 = time.time()
                # IK-goal filter                                                              infer_serial.py:117-129
                volumes = guide.cost(torch.tensor(all_ik_goals.reshape((-1, 7, 1))), 0, batch_size=all_ik_goals.shape[0]).sum(axis=(1, 2)).cpu().numpy()
                indices = np.argsort(volumes)
                goal_joints = all_ik_goals[indices][volumes[indices] < np.min(volumes) + 0.0008]
                goal_joints = goal_joints[np.argmin(np.linalg.norm(start_joints - goal_joints, axis=1))]
                t2 = time.time()
                # the feeder thread draws every scene's whole stream in scene order from the global RandomState, so every scene sees the
                # numbers the reference's loop would give it (nothing else may draw from the global state while a run is in progress).
                # (pool.submit hands the scenes to the lanes in order, the feeder hands the streams out in the same order)
                # where a scene's "Planning Time" (infer_serial.py:108-157: guide construction + IK filter + sampling + best pick) goes
                meta = dict(scene_type=scene_type, scene_num=scene_num, timings=dict(guide_ctor_s=t1 - t0, ik_filter_s=t2 - t1))
                if k == 1:
                    collect(plan(lane, guide, start_joints, goal_joints, feeder.next(), meta, t0))  # serial: the reference's order of events
                else:
                    pending.append(pool.submit(plan, lane, guide, start_joints, goal_joints, feeder.next(), meta, t0))
            while pending:
                collect(pending.pop(0))
    finally:
        if feeder is not None:
            feeder.close()
    if world > 1 or verbose:
        summary = job_summary(results, world)
        if verbose and rank == 0:
            print(f"[infer_serial] {summary['scenes']} scenes on {summary['ranks']} rank(s): success (proxy, collision-free) {summary['success_proxy']}/{summary['scenes']}, "
                  f"strict {summary['success_strict']}/{summary['scenes']}, rows collision-free {summary['rows_collision_free']}/{summary['rows']}")
        run.last_summary = summary
    return results


def main(argv=None):
    parser = argparse.ArgumentParser(prog="Benchmarking Diffusion", description="Benchmarking with IK on Test sets")
    parser.add_argument("-c", "--cfg_path", type=str, default="./configs/cfg_c1_plumbing.yaml")
    parser.add_argument("--scenes-in-flight", type=int, default=1, help="plan this many scenes concurrently on the GPU (extension; the reference is serial)")
    parser.add_argument("--max-scenes", type=int, default=None, help="stop after this many scenes of the cfg's order (all ranks together)")
    parser.add_argument("--seed", type=int, default=None, help="np.random.seed(seed + rank) before the loop (the reference never seeds; for repeatable runs)")
    parser.add_argument("--results-json", type=str, default=None, help="write this rank's per-scene results (without the trajectories) and the job summary "
                                                                       "to PATH (rank 0) / PATH.rank<r> (other ranks)")
    args = parser.parse_args(argv)
    rank = int(os.environ.get("RANK", "0"))
    if args.seed is not None:
        np.random.seed(args.seed + rank)
    results = run(args.cfg_path, scenes_in_flight=args.scenes_in_flight, max_scenes=args.max_scenes)
    if args.results_json:
        import json

        rows = [{k: v for k, v in r.items() if k != "trajectory"} for r in results]
        with open(args.results_json + ("" if rank == 0 else f".rank{rank}"), "w") as f:
            json.dump({"rank": rank, "summary": getattr(run, "last_summary", None), "scenes": rows}, f, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))
    return results


if __name__ == "__main__":
    main()
