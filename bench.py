#!/usr/bin/env python
"""bench.py — guided denoise-steps/sec of the EDMP sampler hot path on MI355X.

One "step" = one full Diffusion.denoise_guided call (T=255 reverse steps, 125 of them guided) over one batch of
B=1024 synthetic trajectories per GPU + the end-of-sampling best-trajectory selection (and, for N>1, the RCCL gather).
value = trajectories x denoise-steps / second over all ranks.  Inputs (weights, scene tables, the (T+1,B,7,50) f64
noise stream) are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

T, N, C = 255, 50, 7
FULL_DIMS = (32, 64, 128, 256, 512, 512)
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2516.6  # MI355X_MICROARCH.md: dense bf16 MFMA peak (16 x the fp32 rate; no sparsity)
SURVEY_FLOPS_PER_TRAJ_STEP = 187_339_904


def effective_cores() -> int:
    """host cores this process may actually use: min(affinity, cgroup CPU quota).  os.cpu_count() alone over-counts
    inside a quota-limited container and oversubscribed torch threads are orders of magnitude slower."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, n)


def pmc_traffic():
    """HBM MB per conv-family launch from the committed rocprofv3 PMC passes (profiles/r06_pmc_hbm_traffic.json, else
    an earlier round's; produced by scripts/pmc_unet_forward.py + scripts/summarize_pmc.py).  Counters cannot be read live."""
    for name in ("r06_pmc_hbm_traffic.json", "r05_pmc_hbm_traffic.json", "r04_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json", "r01_pmc_hbm_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name), "rb") as f:
                raw = f.read()
            d = json.loads(raw)["conv_family"]
            import hashlib

            blob = hashlib.sha1(b"blob %d\0" % len(raw) + raw).hexdigest()  # = `git hash-object`: which committed file this number is
            return {"hbm_MB_per_launch": d["hbm_MB_per_launch"], "hbm_bytes_per_traj_step": d["hbm_bytes_per_traj_step"], "source": f"profiles/{name} (PMC pass, not live)",
                    "source_git_blob": blob}
        except Exception:
            continue
    return None


def traffic_bytes_per_launch(detail):
    """`roofline.traffic` proper: HBM bytes per launch of the dominant kernel family (a number, or None)."""
    return None if not detail else float(detail["hbm_MB_per_launch"]) * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024, help="rows per GPU")
    ap.add_argument("--guides", type=str, default="1,2,3,4,5,10", help="guide ensemble, e.g. 1,2,3 (BASELINE config 2) or 1,2,3,4,5,10,11,13 (config 5)")
    ap.add_argument("--logical-batch", action="store_true",
                    help="BASELINE config 5 semantics: the ranks hold row shards of ONE reference batch of gpus x batch rows; sum(g^2) is all-reduced "
                         "over RCCL once per guided step from inside the device-resident loop (instead of independent replicas)")
    ap.add_argument("--hook", choices=("native", "python"), default="native",
                    help="--logical-batch: the per-guided-step all-reduce as native code (ncclAllReduce called by the device loop, csrc/rccl_hook.hip; "
                         "needs the nccl backend) or as the Python callback into torch.distributed (round 5; the only choice over gloo)")
    ap.add_argument("--obstacles", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the instrumented (HIP-event) pass")
    ap.add_argument("--cpu-steps", type=int, default=16, help="reverse steps of the bounded CPU-baseline sample")
    ap.add_argument("--no-two-scenes", action="store_true", help="skip the informative two-scenes-in-flight measurement")
    ap.add_argument("--no-native-leg", action="store_true", help="skip the A/B leg with every conv on the fp32-MFMA kernels (value_native_f32)")
    ap.add_argument("--no-problem-set", action="store_true", help="skip the informative problem-set measurement (16 distinct scenes through infer_serial.run)")
    args = ap.parse_args()

    backend = os.environ.get("EDMP_DIST_BACKEND", "nccl")  # "gloo" lets N ranks share one GPU (single-GPU test boxes)
    ngpu = torch.cuda.device_count()
    if args.gpus < 1:
        raise SystemExit(f"--gpus {args.gpus}: need at least one rank")
    if args.gpus > 1 and backend == "nccl" and ngpu < args.gpus:
        # RCCL refuses two ranks on one device; a run that silently used fewer GPUs would print a mislabelled line
        raise SystemExit(f"--gpus {args.gpus} but only {ngpu} GPU(s) visible (EDMP_DIST_BACKEND=gloo lets ranks share a GPU: test boxes only)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N` (the reference's launch model is one process per device,
        # benchmark/cfgs/cfg2.yaml:2,14): re-exec under the launcher the contract names instead of falling through to one rank
        import socket

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__), *sys.argv[1:]]
        print(f"[bench] --gpus {args.gpus} without a launcher: re-exec as {' '.join(cmd[1:9])} ...", file=sys.stderr, flush=True)
        os.execv(sys.executable, cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev_index = local_rank if backend == "nccl" else local_rank % max(ngpu, 1)
    torch.cuda.set_device(dev_index)
    dev = f"cuda:{dev_index}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend)
    elif args.logical_batch:
        # a world of one still runs the per-guided-step RCCL all-reduce: N = 1 then measures the hook + collective launch cost
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend, rank=0, world_size=1, **({"device_id": torch.device(dev)} if backend == "nccl" else {}))

    from edmp_amd import dist as ED
    from edmp_amd import guide_cfg as GC
    from edmp_amd import scenes
    from edmp_amd.diffusion import Diffusion
    from edmp_amd.guide import IntersectionVolumeGuide
    from edmp_amd.temporalunet import TemporalUNet

    B = args.batch
    guides = [int(g) for g in args.guides.split(",")]
    logical = bool(args.logical_batch)
    if logical:
        # one reference batch of world*B rows, guide g owning a contiguous row block; rank r holds rows [r*B, (r+1)*B)
        full = GC.build_guide_cfgs([GC.catalog_guide_dict(g) for g in guides], 0, T, rows_per_guide=GC.split_rows(world * B, len(guides)))
        cfgs = ED.shard_guide_cfgs(full, rank * B, (rank + 1) * B)
        scene = scenes.random_scene(11, args.obstacles)  # one planning problem for the whole batch
    else:
        cfgs = GC.build_guide_cfgs([GC.catalog_guide_dict(g) for g in guides], 0, T, rows_per_guide=GC.split_rows(B, len(guides)))
        scene = scenes.random_scene(11 + rank, args.obstacles)  # every rank = its own planning problem replica
    start, goal = scenes.DEFAULT_START, scenes.DEFAULT_GOAL

    net = TemporalUNet(None, C, 32, dev, dims=FULL_DIMS, seed=1, max_batch=B)
    guide = IntersectionVolumeGuide(scene, dev, cfgs, B)
    dif = Diffusion(T, dev)
    ctx = dif.ctx
    noise_host = np.random.RandomState(1234 + rank).standard_normal((T + 1, B, C, N))
    noise = ctx.to_dev(noise_host, torch.float64)
    ctx.sync()

    import functools

    # logical-batch mode issues the collective even in a world of one, so that N = 1 measures the hook + RCCL launch cost
    ar = None
    if logical and args.hook == "native" and backend == "nccl":
        ar = ED.RcclAllReduce(dif)  # own communicator: the unique id travels through the process group once, before the timed region
    elif logical:
        ar = functools.partial(ED.allreduce_sum_, always=(world > 1 or (dist.is_available() and dist.is_initialized())))

    last = {}

    def one_call():
        X = dif.denoise_guided(net, guide, N, C, cfgs["guidance_schedule"], batch_size=B, start=start, goal=goal, noise=noise, return_device=True,
                               allreduce=ar, zero_row0=(rank == 0 or not logical))
        vols, idx = guide.row_swept_volumes(start, goal, X)  # synchronises (argmin comes back to the host)
        # plan success of EVERY row (exact link-box vs obstacle check, csrc/success.hip): the second half of the metric
        sr = guide.success_rows(X)
        last["X"] = X
        traj = X[idx].cpu().numpy()
        res = ED.gather_best(float(vols[idx]), idx, traj, bool(sr["ok"][idx]), device=dev, rows_ok=sr["rows_ok"], rows=sr["rows"],
                             collision_free=bool(sr["collision_free"][idx]), rows_collision_free=sr["rows_collision_free"])
        res["aabb_volume_zero"] = ED.geometric_success(float(vols[idx]), traj)  # the guide's own (conservative) criterion, rank-local
        return res

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        best = one_call()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        best = one_call()
    barrier()
    dt = time.perf_counter() - t0
    n_ranks_seen = 1
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # the world size the collective library actually ran with: every rank contributes 1 to a SUM
        ones = torch.ones(1, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        n_ranks_seen = int(round(float(ones.item())))
        if n_ranks_seen != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group summed {n_ranks_seen} ranks")
    value = world * B * T * args.steps / dt

    # ---- A/B leg (round 6): the same workload with EVERY conv on the fp32-MFMA kernels (EDMP_BF16X3=0 at model-build time) --------
    # `value` is the default product path: 26 of the 40 conv launches of a reverse step form each fp32 product as six exact bf16
    # partial products on the bf16 matrix pipe, fp32 accumulation (csrc/bf3.hip) - measured at HALF the fp32-MFMA kernels' error
    # against float64 (profiles/r06_bf16x3.md, tests: test_bf16x3_split_layers_are_at_least_as_accurate_as_the_fp32_mfma_layers).  The native leg keeps the two numbers side by side.
    native = None
    if not args.no_native_leg:
        env_prev = os.environ.get("EDMP_BF16X3")
        os.environ["EDMP_BF16X3"] = "0"
        try:
            net_n = TemporalUNet(None, C, 32, dev, dims=FULL_DIMS, seed=1, max_batch=B)
        finally:
            if env_prev is None:
                del os.environ["EDMP_BF16X3"]
            else:
                os.environ["EDMP_BF16X3"] = env_prev
        net_default, n_steps_native = net, max(1, min(args.steps, 5))
        net = net_n
        try:
            one_call()
            barrier()
            t0 = time.perf_counter()
            for _ in range(n_steps_native):
                one_call()
            barrier()
            dtn = time.perf_counter() - t0
        finally:
            net = net_default
        if world > 1:
            ttn = torch.tensor([dtn], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(ttn, op=dist.ReduceOp.MAX)
            dtn = float(ttn.item())
        native = {"value": world * B * T * n_steps_native / dtn, "ms_per_step": 1e3 * dtn / n_steps_native, "steps": n_steps_native, "warmup": 1}
        del net_n
        one_call()  # re-bind the default model (resident slot): the instrumented passes below read ITS layer program

    out = None
    if rank == 0:
        nominal, executed = net.flops_per_trajectory()
        f32_bf16 = net.flops_by_pipe()
        out = {
            "metric": "guided denoise-steps/sec (trajectories x reverse steps)",
            "value": value,
            "unit": "traj-steps/s",
            "n_gpus": world,
            "n_ranks_seen": n_ranks_seen,  # summed over the process group (RCCL / gloo) inside the run: equals n_gpus or the run aborts
            "dist_backend": (backend if world > 1 else None),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "dtype_variant": (None if f32_bf16[1] == 0 else "f32 results, fp32 accumulation; 26 of the 40 conv launches of a reverse step (the 20 Conv1dBlocks at L = 13 / 7 / 4 and the six k3s2 / ConvTranspose "
                              "resamplers of the >= 128-channel levels) form every fp32 product as six EXACT bf16 x bf16 partial products on the bf16 matrix pipe (bf16x3 split, "
                              "csrc/bf3.hip); error vs float64 0.43-0.62 x (rmse) and <= 1.03 x (max) the fp32-MFMA kernels' (profiles/r06_bf16x3.md); EDMP_BF16X3=0 selects the "
                              "all-fp32-MFMA program = value_native_f32"),
            "value_bf16x3": (value if f32_bf16[1] > 0 else None),
            "value_native_f32": (value if f32_bf16[1] == 0 else (native["value"] if native else None)),
            "native_f32_leg": native,
            "data": "synthetic",
            "config": {
                "workload": f"denoise_guided T={T} N={N} batch={B}/GPU, {len(guides)}-guide ensemble {guides}, {args.obstacles}-cuboid synthetic scene, full TemporalUNet (29.9M params, random init), f64 state / f32 denoiser+guide",
                "global_batch": world * B,
                "parallelism": (f"one logical batch of {world * B} rows row-sharded x{world}: RCCL all-reduce of sum(g^2) per guided step inside the device loop + end-of-sampling gather"
                                if logical else (f"row-sharded replicas x{world}, end-of-sampling RCCL gather" if world > 1 else "single GPU")),
            },
            "best": {"rank": best["rank"], "row": best["index"], "swept_volume": best["volume"], "aabb_volume_zero_and_within_limits": best["aabb_volume_zero"]},
            "success_proxy": {"rows_collision_free": best["rows_collision_free"], "rows": best["rows"],
                              "collision_free_rate": best["rows_collision_free"] / max(best["rows"], 1), "best_row_collision_free": best["collision_free"],
                              "rows_ok": best["rows_ok"], "rate": best["rows_ok"] / max(best["rows"], 1), "best_row_ok": best["success"],
                              "note": "plan success-rate half of the metric, computed INSIDE the timed call for every row of the batch (summed over ranks).  "
                                      "collision_free_rate / best_row_collision_free = the REFERENCE's criterion: no link box meets an obstacle (exact oriented-box / "
                                      "cylinder test, 4 interpolated configurations per segment); leaving the joint limits only prints there (lib/environment.py:659-661, "
                                      "672).  rate / best_row_ok = the stricter flag that also requires every waypoint inside the limits.  A geometric stand-in for the "
                                      "reference's pybullet check (lib/environment.py:632-680), which is unavailable offline; random-init denoiser (no trained weights "
                                      "offline): the rates say nothing about planning quality, they are reported, never gated"},
            # one logical batch: host time of the per-guided-step hook that enqueues the RCCL all-reduce of sum(g^2) on the context's
            # stream, measured by the library around each call (edmp_sampler_allreduce_stats), rank 0, last call
            "allreduce_hook": (None if not logical else {"kind": dif.hook_stats["kind"], "calls_per_denoise": dif.hook_stats["calls"],
                                                        "avg_us": 1e6 * dif.hook_stats["total_s"] / max(dif.hook_stats["calls"], 1), "max_us": 1e6 * dif.hook_stats["max_s"]}),
            "unet_flops_per_traj_step": {"nominal": nominal, "direct_form_after_tap_skipping": net.flops_direct_form(), "issued_mfma_fp32_equivalent": executed,
                                         "issued_on_the_fp32_pipe": f32_bf16[0], "issued_on_the_bf16_pipe": f32_bf16[1], "survey": SURVEY_FLOPS_PER_TRAJ_STEP,
                                         "note": "issued (fp32 equivalent) < direct form: the L=2 / L=4 Conv1dBlocks run in Karatsuba form (3 of 4 resp. 9 of 14 matrix products); the bf16x3 layers "
                                                 "issue 6 bf16 MFMA FLOPs per fp32 FLOP they replace, on the bf16 pipe"},
        }

    # ---- cost of the success kernel: on the batch just produced (random-init rows collide early and leave the obstacle loop at the
    # first hit: the EASY case) and on a collision-free in-limit batch against the scene moved 10 m away (every configuration x link
    # box x obstacle tested: the WORST case, what trained weights should produce) ------------------------------------------------
    if world == 1 and rank == 0:
        from edmp_amd import franka

        def timed_check(g_, X_):
            g_.success_rows(X_, return_device=True)
            best_ms = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(ctx.stream)
                r_ = g_.success_rows(X_, return_device=True)
                e1.record(ctx.stream)
                e1.synchronize()
                best_ms = min(best_ms, e0.elapsed_time(e1))
            return best_ms, r_

        ms_this, _ = timed_check(guide, last["X"])
        far = scene.copy()
        far[:, 0] += 10.0
        kinds = np.zeros(far.shape[0], dtype=np.int32)
        kinds[:: 5] = 1  # every fifth obstacle a true cylinder of radius dims[0] (dims = (r, r, h), datasets/load_test_dataset.py:136-139)
        far[kinds == 1, 8] = far[kinds == 1, 7]
        lo_q, hi_q = franka.joint_limits()
        rs_ = np.random.RandomState(5)
        qa, qb = rs_.uniform(lo_q + 0.05, hi_q - 0.05, (B, C)), rs_.uniform(lo_q + 0.05, hi_q - 0.05, (B, C))
        tt_ = np.linspace(0.0, 1.0, N)
        Xfree = ctx.to_dev(qa[:, :, None] * (1 - tt_) + qb[:, :, None] * tt_, torch.float64)
        gfar = IntersectionVolumeGuide(far, dev, cfgs, B, obstacle_kinds=kinds)
        ms_free, r_free = timed_check(gfar, Xfree)
        guide.success_rows(last["X"])  # re-bind the bench scene
        out["success_proxy"]["check_ms"] = {"this_batch": ms_this, "collision_free_batch_worst_case": ms_free, "collision_free_rows_of_that_batch": r_free["rows_collision_free"],
                                            "share_of_step_worst_case": ms_free / (1e3 * dt / args.steps),
                                            "note": "edmp_success_rows_dev incl. the flag count and its 16-byte read-back, HIP events on the context's stream, best of 3"}
        del gfar, Xfree

    # ---- informative: whole-scene wall time when the noise is NOT pre-resident (never `value`) -------------------------
    if world == 1 and rank == 0 and not logical:
        def scene_time(**kw):
            t1 = time.perf_counter()
            X = dif.denoise_guided(net, guide, N, C, cfgs["guidance_schedule"], batch_size=B, start=start, goal=goal, return_device=True, **kw)
            guide.row_swept_volumes(start, goal, X)
            torch.cuda.synchronize()
            return time.perf_counter() - t1

        from edmp_amd import nprng

        np.random.seed(0)
        t_np_first = scene_time()                # reference contract: NumPy draw (91 M normals) + 734 MB upload + loop; the
        t_np = min(scene_time(), scene_time())   # first scene of a process also creates the draw thread and its OpenMP team
        t_dev = scene_time(noise="device", seed=1)  # non-parity mode: Philox on the GPU
        out["end_to_end_scene_seconds"] = {
            "noise_resident_in_hbm": dt / args.steps,
            "numpy_stream_drawn_and_uploaded_per_scene": t_np,
            "numpy_stream_first_scene_of_the_process": t_np_first,
            "device_philox_noise": t_dev,
            "traj_steps_per_s": {"numpy_stream": B * T / t_np, "device_noise": B * T / t_dev},
            "host_draw": {"threads_and_cache_domain_cores": list(nprng.team(nprng.draw_threads())),
                          "note": "NumPy's legacy RandomState stream reproduced bit for bit (edmp_amd/nprng.py) by a team confined to one last-level-cache domain, "
                                  "drawn into pinned memory in chunks of 1, 2, 4, 8, 16, 16, ... steps and uploaded by DMA beside the kernels"},
        }

    # ---- informative: TWO independent scenes in flight on this GPU (never `value`: the named config is one batch of 1024) ----
    # Every launch of the loop is one wave of 256 workgroups, so one chain leaves the chip idle in each dispatch gap and
    # under-filled in each kernel tail; a second, independent scene on its own context / stream / host thread fills the holes.
    # Throughput mode for a driver that has many scenes to plan (the reference's scene loop, infer_serial.py:95-170, is serial).
    if world == 1 and rank == 0 and not logical and not args.no_two_scenes:
        import threading

        from edmp_amd.runtime import lane_context

        ctx2 = lane_context(dev_index, 1)
        net2 = TemporalUNet(None, C, 32, ctx2, dims=FULL_DIMS, seed=1, max_batch=B)
        guide2 = IntersectionVolumeGuide(scenes.random_scene(12, args.obstacles), ctx2, cfgs, B)
        dif2 = Diffusion(T, ctx2)
        noise2 = ctx2.to_dev(np.random.RandomState(4321).standard_normal((T + 1, B, C, N)), torch.float64)
        ctx2.sync()

        chain_errors = []

        def chain(d, n_, g_, z_, k):
            try:
                for _ in range(k):
                    X = d.denoise_guided(n_, g_, N, C, cfgs["guidance_schedule"], batch_size=B, start=start, goal=goal, noise=z_, return_device=True)
                    g_.row_swept_volumes(start, goal, X)
                    g_.success_rows(X)
            except BaseException as exc:  # a thread must not fail silently: the pair time would be meaningless
                chain_errors.append(exc)

        chain(dif2, net2, guide2, noise2, 1)
        torch.cuda.synchronize()
        k2 = max(2, args.steps)
        t1 = time.perf_counter()
        ths = [threading.Thread(target=chain, args=a) for a in ((dif, net, guide, noise, k2), (dif2, net2, guide2, noise2, k2))]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        torch.cuda.synchronize()
        t_pair = (time.perf_counter() - t1) / k2
        if chain_errors:
            raise chain_errors[0]
        out["two_scenes_in_flight"] = {"traj_steps_per_s": 2 * B * T / t_pair, "ms_per_pair_of_scenes": 1e3 * t_pair, "vs_value": (2 * B * T / t_pair) / value,
                                       "note": "two independent 1024-row scenes on two contexts (streams) of this GPU, one host thread each; results bit-identical to the "
                                               "one-at-a-time runs; informative, not the named config (one batch of 1024)"}
        del net2, guide2, dif2, noise2

    # ---- informative: a PROBLEM SET the way the reference counts a scene (never `value`) --------------------------------------
    # The reference's per-scene clock (infer_serial.py:108-157) covers guide construction + IK-goal filter + sampling + best pick;
    # the timed loop above builds the guide once.  infer_serial.run over 16 distinct synthetic scenes of this workload's size
    # (true cylinders included, noise drawn per scene from NumPy's global RandomState), serial and with two scenes in flight.
    if world == 1 and rank == 0 and not logical and not args.no_problem_set:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import problem_set_bench

        try:  # informative leg: a failure here (pinned allocation, config) must not lose the measured line
            out["problem_set"] = problem_set_bench.measure(16, B, tuple(guides), args.obstacles, min(3, args.obstacles), device=dev)
        except Exception as exc:  # noqa: BLE001
            out["problem_set"] = {"error": f"{type(exc).__name__}: {exc}"}

    # ---- roofline of the dominant kernel family (fp32-MFMA conv kernels of the UNet), N=1 only ----------------------
    # Two extra, instrumented calls with HIP events on the context's stream:
    #   pass A (edmp_prof_enable 2): ONE event pair around the whole UNet layer program of every reverse step -> the conv
    #          family's time per call with negligible overhead (510 events per call); like rocprofv3's back-to-back kernel
    #          trace it contains the dispatch gaps between the program's dependent launches.  conv_ms <= ms_per_step.
    #   pass B (edmp_prof_enable 1): one pair around EVERY conv launch, read per program op (edmp_prof_ops).  The ~26 k
    #          extra events cost several us per launch, so pass B only supplies each kernel's share: the measured
    #          per-launch overhead (sum of pass B - pass A, divided by the launches) is subtracted from every bracket.
    # FLOPs are the EXECUTED ones (taps that fall into the zero padding are never issued); the nominal SURVEY 8(d)
    # figure is reported next to it.
    if world == 1 and rank == 0 and not args.no_roofline:
        ctx.prof(2)
        ctx.prof_read(reset=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        one_call()
        torch.cuda.synchronize()
        passA_wall_ms = 1e3 * (time.perf_counter() - t1)
        conv_ms, launches = ctx.prof_read(reset=True)
        ctx.prof(1)
        one_call()
        ops = ctx.prof_ops()
        ops_bf16 = ctx.prof_ops_bf16()
        ev_ms, launches_b = ctx.prof_read(reset=True)
        ctx.prof(0)
        ms_step = 1e3 * dt / args.steps
        over_us = 1e3 * (ev_ms - conv_ms) / max(launches_b, 1)  # event overhead per bracketed launch in pass B
        nominal, executed = net.flops_per_trajectory()
        f32_issued, bf16_issued = net.flops_by_pipe()
        head = 2.0 * N * C * FULL_DIMS[0]  # the 1x1 head runs inside the (VALU) posterior kernel
        conv_nominal, conv_exec = nominal - head, executed - head
        f32_issued -= head
        table = {}
        for (name, calls, ms, fl), fb in zip(ops, ops_bf16):
            if calls == 0:
                continue
            r = table.setdefault(name, {"launches": 0, "ms": 0.0, "flop": 0.0, "flop_bf16": 0.0})
            r["launches"] += calls
            r["ms"] += max(ms - 1e-3 * over_us * calls, 0.0)
            r["flop"] += fl * B * calls
            r["flop_bf16"] += fb * B * calls
        rows = []
        for name, r in sorted(table.items(), key=lambda kv: -kv[1]["ms"]):
            tf = r["flop"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0.0
            row = {"kernel": name, "launches": r["launches"], "avg_us": 1e3 * r["ms"] / r["launches"], "share": r["ms"] / conv_ms if conv_ms else 0.0}
            if r["flop_bf16"] > 0:  # a bf16x3 kernel: its matrix work is issued on the bf16 pipe; `executed_tflops` stays the fp32 work it REPLACES
                tb16 = r["flop_bf16"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0.0
                row.update({"pipe": "bf16", "issued_tflops": tb16, "frac": tb16 / PEAK_BF16_MFMA_TFLOPS, "fp32_equivalent_tflops": tf, "fp32_equivalent_vs_fp32_peak": tf / PEAK_F32_MFMA_TFLOPS})
            else:
                row.update({"pipe": "f32", "executed_tflops": tf, "frac": tf / PEAK_F32_MFMA_TFLOPS})
            rows.append(row)
        conv_direct = net.flops_direct_form() - head
        t_conv = conv_ms * 1e-3
        # matrix-pipe time at peak of one call: every kernel's issued work on ITS pipe at that pipe's dense peak
        pipe_s = (f32_issued / (PEAK_F32_MFMA_TFLOPS * 1e12) + bf16_issued / (PEAK_BF16_MFMA_TFLOPS * 1e12)) * B * T
        issued_total = (f32_issued + bf16_issued) * B * T
        ach = issued_total / t_conv / 1e12
        peak_eff = issued_total / pipe_s / 1e12  # the rate this mix of fp32- and bf16-pipe work would run at if every kernel sat on its roof
        ach_direct = conv_direct * B * T / t_conv / 1e12
        tb = traffic_bytes_per_launch(pmc_traffic())
        avg_s = 1e-3 * conv_ms / max(launches, 1)
        out["roofline"] = {
            "kernel": "MFMA conv family of the TemporalUNet: edmp::wide_conv_kernel (position-tile Conv1d k5 + GroupNorm + Mish, k3s2, ConvTranspose k4s2; fp32 MFMA, Karatsuba forms at L=2/4) + "
                      "edmp::bf3_conv_kernel (the same op at L=13 / 7 / 4 and the resamplers as six exact bf16 partial products per fp32 product on the bf16 pipe, fp32 accumulation) + "
                      "edmp::level_kernel (whole 32/64-channel levels, fp32 MFMA)",
            "bound": "mfma",
            "achieved": ach,
            "peak": peak_eff,
            "unit": "TFLOP/s",
            "frac": ach / peak_eff,
            "flops": "MFMA work actually ISSUED, each kernel on its own pipe (fp32: padding taps never issued, Karatsuba forms at L=2/L=4 issue 3 of 4 / 9 of 14 products; bf16x3 kernels: 6 x the "
                     "direct form, on the bf16 pipe).  peak = issued work / (fp32-pipe work / 157.3 + bf16-pipe work / 2516.6 TFLOP/s): frac = matrix-pipe time at peak / measured time",
            "pipes": {"f32": {"issued_flop_per_traj_step": f32_issued, "peak_tflops": PEAK_F32_MFMA_TFLOPS, "pipe_ms_at_peak_per_call": 1e3 * f32_issued * B * T / (PEAK_F32_MFMA_TFLOPS * 1e12)},
                      "bf16": {"issued_flop_per_traj_step": bf16_issued, "peak_tflops": PEAK_BF16_MFMA_TFLOPS, "pipe_ms_at_peak_per_call": 1e3 * bf16_issued * B * T / (PEAK_BF16_MFMA_TFLOPS * 1e12)}},
            # the same time against the fp32 work the program REPLACES (direct form, padding taps skipped) and the fp32 pipe's peak: what an all-fp32-MFMA program would need to reach
            "achieved_direct_form": ach_direct,
            "frac_direct_form": ach_direct / PEAK_F32_MFMA_TFLOPS,
            "achieved_fp32_equivalent_issued": conv_exec * B * T / t_conv / 1e12,
            "frac_fp32_equivalent_issued_vs_fp32_peak": conv_exec * B * T / t_conv / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "traffic": tb,  # HBM bytes per conv launch (PMC FETCH_SIZE x2 + WRITE_SIZE passes, committed under profiles/)
            "traffic_detail": pmc_traffic(),
            # north-star asks for the HBM fraction too: PMC bytes per launch / average launch duration vs 8 TB/s
            "hbm": None if tb is None else {"achieved": tb / avg_s / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": tb / avg_s / 1e9 / 8000.0},
            "achieved_nominal": conv_nominal * B * T / t_conv / 1e12,
            "launches": launches,
            "avg_launch_us": 1e3 * conv_ms / max(launches, 1),
            "flops_per_launch_issued": issued_total / max(launches, 1),
            "conv_ms_per_call": conv_ms,
            "conv_share_of_step": conv_ms / ms_step,
            "timing": {"ms_per_step_timed": ms_step, "passA_program_brackets_conv_ms": conv_ms, "passA_wall_ms": passA_wall_ms, "passB_per_launch_brackets_ms": ev_ms,
                       "passB_event_overhead_us_per_launch": over_us},
            "per_kernel": rows,
            "note": "achieved = issued MFMA FLOPs of one denoise_guided call / conv_ms_per_call (one HIP-event pair around each reverse step's layer program: "
                    "includes the inter-kernel dispatch gaps, like rocprofv3's back-to-back kernel trace); per_kernel rows = per-launch brackets minus the measured "
                    "event overhead per launch, each against its own pipe's peak; achieved_nominal counts every tap (SURVEY 8d)",
        }

    # ---- CPU baseline: the oracle (port of the reference) on this box's host cores, bounded sample -----------------
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        from edmp_amd import weights as W
        from oracle import edmp_oracle as O

        cores = effective_cores()
        torch.set_num_threads(cores)
        sd = W.init_state_dict(1, C, 32, FULL_DIMS)
        om, og = O.UNetOracle(sd), O.GuideOracle(scene, cfgs, B)
        k = args.cpu_steps
        small = noise_host[: k + 1]
        t0 = time.perf_counter()
        O.denoise_guided(om, og, T, N, C, cfgs["guidance_schedule"], B, start, goal, noise=small, t_stop=T - k)  # the oracle reads noise[0..k] only
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": B * k / cdt,
            "unit": "traj-steps/s",
            "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"oracle (NumPy f64 + torch-CPU f32 restatement of the reference, no_grad) on the same workload, reverse steps t={T}..{T - k + 1} ({k} of {T} steps, {k // 2} guided), {cdt:.1f} s",
        }
    if rank == 0:
        print(json.dumps(out))
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
