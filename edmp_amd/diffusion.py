"""Diffusion — host-side mirror of the reference sampler object (diffusion/diffusion.py:8-356).  The reverse loop
(`denoise_guided`) runs device-resident in libedmp_hip.so (edmp_amd/csrc/sampler.hip): one host call per scene."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _capi, franka, nprng
from .runtime import get_context, ptr


def draw_noise(T: int, batch_size: int, num_channels: int, traj_len: int) -> np.ndarray:
    """(T+1, B, C, N) f64 from the GLOBAL NumPy RandomState in the reference's call order: one
    ``multivariate_normal(0, I_N, size=(B, C))`` for X_T (diffusion.py:303) then one per step (diffusion.py:126).
    With an identity covariance that call consumes the stream exactly like ``standard_normal((B, C, N))``
    (pinned by tests/test_host.py), so all draws are made in one vectorised call - by edmp_amd.nprng, which produces
    NumPy's legacy stream bit for bit on all host cores and advances the global state exactly like NumPy."""
    return nprng.standard_normal((T + 1, batch_size, num_channels, traj_len))


def _startgoal(start, goal, needed: bool):
    """(7,) f64 start / goal for the C ABI (which reads 7 doubles from each).  The reference allows None when it neither
    conditions nor guides (diffusion.py:253, 300): zeros stand in there; anything else must have exactly 7 entries."""
    out = []
    for name, v in (("start", start), ("goal", goal)):
        if v is None:
            if needed:
                raise ValueError(f"{name} is required when conditioning or guiding")
            v = np.zeros(7)
        a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1))
        if a.size != 7:
            raise ValueError(f"{name} must have 7 joint values, got shape {np.shape(v)}")
        out.append(a)
    return out


class PinnedNoiseStream:
    """A (T+1, B, C, N) f64 noise stream in page-locked host memory that is still BEING DRAWN: the producer (infer_serial's feeder
    thread) fills it front to back and publishes how far it got; `Diffusion.denoise_guided(noise=stream)` uploads chunk after chunk as
    soon as each is complete - the first scene of a run starts after one step's worth of draws, later scenes find their stream ready."""

    def __init__(self, tensor):
        import threading

        self.tensor, self.drawn, self.error = tensor, 0, None
        self._cv = threading.Condition()

    def publish(self, n_doubles, error=None):
        with self._cv:
            self.drawn, self.error = int(n_doubles), error
            self._cv.notify_all()

    def wait_until(self, n_doubles):
        with self._cv:
            while self.drawn < n_doubles and self.error is None:
                self._cv.wait()
            if self.error is not None:
                raise self.error


class Diffusion:
    """Same constructor / method signatures as the reference ``Diffusion(T, device, variance_thresh=0.02)``."""

    def __init__(self, T, device, variance_thresh=0.02):
        self.T = int(T)
        self.variance_thresh = float(variance_thresh)
        self.ctx = get_context(device)
        self.device = self.ctx.device
        self.ctx.ensure_sampler(self.T, self.variance_thresh)
        self.beta = np.zeros(self.T)
        self.alpha = np.zeros(self.T)
        self.alpha_bar = np.zeros(self.T)
        _capi.check(self.ctx.lib.edmp_sampler_read_schedule(self.ctx.h, _capi.as_pd(self.beta), _capi.as_pd(self.alpha), _capi.as_pd(self.alpha_bar)))

    def schedule_variance(self, thresh=0.02):
        return self.beta.copy()

    # ---- single pieces (reference API) ------------------------------------------------------------------------
    def p_sample_using_posterior(self, xt, t, eps, z=None):
        """diffusion.py:116-135.  Draws z from the global NumPy RNG like the reference unless ``z`` is given."""
        ctx = self.ctx
        ctx.ensure_sampler(self.T, self.variance_thresh)
        b, c, n = xt.shape
        if z is None:
            z = nprng.standard_normal((b, c, n))
        X = ctx.to_dev(np.array(xt, dtype=np.float64), torch.float64)  # a fresh device tensor: updated in place below
        e = ctx.to_dev(eps, torch.float32)
        zd = ctx.to_dev(np.asarray(z, dtype=np.float64), torch.float64)
        _capi.check(ctx.lib.edmp_psample_dev(ctx.h, ptr(X), ptr(e), ptr(zd), b, c, n, int(t), 1), "edmp_psample_dev")
        return ctx.to_host(X)

    # ---- forward process (training-side data generation) -------------------------------------------------------
    def _q(self, x, t, eps, cumulative, condition=False):
        ctx = self.ctx
        ctx.ensure_sampler(self.T, self.variance_thresh)
        x = np.ascontiguousarray(x, dtype=np.float64)
        b, c, n = x.shape
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(t), (b,)), dtype=np.int32)
        if eps is None:  # diffusion.py:68-71 / 95-98: an identity-covariance multivariate normal == standard normal draws
            eps = nprng.standard_normal((b, c * n)).reshape(b, c, n)
        xd = ctx.to_dev(x, torch.float64)
        ed = ctx.to_dev(np.ascontiguousarray(eps, dtype=np.float64), torch.float64)
        xt = ctx.empty((b, c, n), torch.float64)
        mean = ctx.empty((b, c, n), torch.float64)
        _capi.check(ctx.lib.edmp_q_sample_dev(ctx.h, ptr(xd), ptr(ed), _capi.as_pi32(t), b, c, n, int(cumulative), int(bool(condition)), ptr(xt),
                                              ptr(mean)), "edmp_q_sample_dev")
        return ctx.to_host(xt), ctx.to_host(mean), t

    def q_sample(self, x, t, eps=None):
        """q(x_t | x_{t-1}) (diffusion.py:52-77): (xt, mean, var)."""
        xt, mean, t = self._q(x, t, eps, cumulative=0)
        return xt, mean, np.sqrt(1 - self.alpha[t - 1])

    def q_sample_from_x0(self, x0, t, eps=None):
        """q(x_t | x_0) (diffusion.py:79-105): (xt, mean, var), var of shape (b,1,1) like the reference."""
        xt, mean, t = self._q(x0, t, eps, cumulative=1)
        return xt, mean, np.sqrt(1 - self.alpha_bar[t - 1, np.newaxis, np.newaxis])

    def generate_q_sample(self, x0, time_steps=None, condition=True, return_type="tensor"):
        """Training pairs (diffusion.py:201-251): random timesteps + noise from the global NumPy RNG in the reference's
        order, diffusion and conditioning on the GPU.  Returns (X, Y, time_steps, means, vars)."""
        b, c, n = x0.shape
        if time_steps is None:
            time_steps = np.random.randint(1, self.T + 1, size=(b,))
        eps = nprng.standard_normal((b, c, n))  # == multivariate_normal(0, I_n, size=(b, c))   (diffusion.py:231)
        xt, means, t = self._q(x0, time_steps, eps, cumulative=1, condition=condition)
        vars_ = np.sqrt(1 - self.alpha_bar[t - 1, np.newaxis, np.newaxis])
        if return_type == "tensor":
            return torch.tensor(xt, dtype=torch.float32), torch.tensor(eps, dtype=torch.float32), torch.tensor(time_steps, dtype=torch.float32), means, vars_
        if return_type == "numpy":
            return xt, eps.copy(), time_steps, means, vars_
        raise ValueError('return_type must be "tensor" or "numpy"')  # the reference falls through to a NameError here

    def set_graph_replay(self, on: bool):
        """Capture the device-resident loop of denoise_guided into a hipGraph and replay it (see edmp_sampler_set_graph)."""
        self.ctx.ensure_sampler(self.T, self.variance_thresh)
        _capi.check(self.ctx.lib.edmp_sampler_set_graph(self.ctx.h, 1 if on else 0))

    def clip_joints(self, joints):
        lo, hi = franka.joint_limits()
        return np.clip(joints, lo[np.newaxis, :, np.newaxis], hi[np.newaxis, :, np.newaxis])

    # ---- the loop ----------------------------------------------------------------------------------------------
    def _prepare(self, model, guide, batch_size, guidance_schedule):
        ctx = self.ctx
        if model.ctx is not ctx or (guide is not None and guide.ctx is not ctx):
            raise _capi.EdmpError("model, guide and diffuser must live on the same GPU")
        ctx.ensure_sampler(self.T, self.variance_thresh)
        model._bind()
        if guide is not None:
            guide._bind()
            if guide.batch_size != batch_size:
                raise ValueError(f"guide was built for batch {guide.batch_size}, denoise_guided called with {batch_size}")
            guide._set_rows(guidance_schedule if guidance_schedule is not None else guide._sched)

    def denoise_guided(self, model, guide, traj_len, num_channels, guidance_schedule, batch_size=1, start=None, goal=None,
                       condition=True, benchmarking=False, *, noise=None, seed=0, t_stop=0, zero_row0=True, return_device=False, chunk_steps=16, allreduce=None):
        """diffusion.py:300-356.  ``noise``: optional pre-drawn (T+1,B,C,N) f64 ndarray / device tensor (default:
        drawn from the global NumPy RNG in the reference's order); ``noise="device"`` draws z on the GPU (Philox,
        ``seed``) — a non-parity mode without the host draw / upload.  ``allreduce``: this call is one row shard
        of a batch spread over several GPUs; an ``edmp_amd.dist.RcclAllReduce`` (native ncclAllReduce inside the device loop) or a
        callable that sums the f64 device scalar over ranks in place (edmp_amd.dist.allreduce_sum_; a Python callback per guided step).
        Returns (B,C,N) f64 ndarray (a fresh copy)."""
        ctx = self.ctx
        self._prepare(model, guide, batch_size, guidance_schedule)
        _capi.check(ctx.lib.edmp_sampler_set_condition(ctx.h, 1 if condition else 0))
        s, g = _startgoal(start, goal, needed=bool(condition) or guide is not None)
        if int(traj_len) != model.horizon or int(num_channels) != model.input_dim:
            raise ValueError(f"traj_len/num_channels ({traj_len}, {num_channels}) do not match the model's ({model.horizon}, {model.input_dim})")
        if not 0 <= int(t_stop) < self.T:
            raise ValueError(f"t_stop must lie in [0, {self.T}), got {t_stop}")
        out = ctx.empty((batch_size, num_channels, traj_len), torch.float64)
        if allreduce is not None:
            # "one logical batch across GPUs": this rank holds a row shard of a larger reference batch; the only cross-row
            # coupling, the whole-batch sum(g^2) (lib/guide.py:629), is summed over ranks between the two halves of every
            # guided step - INSIDE the device-resident loop: the library calls the hook once per guided step with the
            # context's stream and the device scalar, the collective is ordered by the stream (no host round trip)
            if noise is None or isinstance(noise, str):
                raise ValueError("sharded runs take an explicit noise array (this rank's rows of the global stream)")
            from .dist import RcclAllReduce

            def _read_stats():
                raw = (C.c_uint64 * 3)()
                _capi.check(ctx.lib.edmp_sampler_allreduce_stats(ctx.h, raw, 1))
                # host time inside the hook, measured by the library around each call (any hook): GIL + collective enqueue
                self.hook_stats = dict(calls=int(raw[0]), total_s=1e-9 * int(raw[1]), max_s=1e-9 * int(raw[2]),
                                       kind="native ncclAllReduce (csrc/rccl_hook.hip)" if native else "python callback (ctypes -> torch.distributed)")

            native = isinstance(allreduce, RcclAllReduce)
            if native:
                # the hook is native code installed once on this context (csrc/rccl_hook.hip): nothing to install per call
                if allreduce.ctx is not ctx or not allreduce.attached():
                    raise ValueError("this RcclAllReduce is not attached to the diffuser's context (or was closed)")
                _capi.check(ctx.lib.edmp_rccl_enable(ctx.h, 1), "edmp_rccl_enable")
                _read_stats()
                try:
                    return self.denoise_guided(model, guide, traj_len, num_channels, guidance_schedule, batch_size, start, goal, condition, benchmarking,
                                               noise=noise, seed=seed, t_stop=t_stop, zero_row0=zero_row0, return_device=return_device)
                finally:
                    _capi.check(ctx.lib.edmp_rccl_enable(ctx.h, 0), "edmp_rccl_enable")  # other runs of this context are not shards
                    _read_stats()
            sumsq = self.sumsq_tensor()

            def _hook(_user, _stream, _ptr):
                try:
                    with torch.cuda.stream(ctx.stream):  # RCCL orders itself after the gradient kernels / before step_b
                        allreduce(sumsq)
                    return 0
                except Exception as exc:  # surfaced by the C side as EDMP_ERR_STATE
                    self._hook_error = exc
                    return 1

            cb = _capi.ALLREDUCE_FN(_hook)
            self._hook_error = None
            _capi.check(ctx.lib.edmp_sampler_set_allreduce(ctx.h, C.cast(cb, C.c_void_p), None))
            _read_stats()
            try:
                res = self.denoise_guided(model, guide, traj_len, num_channels, guidance_schedule, batch_size, start, goal, condition, benchmarking,
                                          noise=noise, seed=seed, t_stop=t_stop, zero_row0=zero_row0, return_device=return_device)
            except _capi.EdmpError:
                if self._hook_error is not None:
                    raise self._hook_error
                raise
            finally:
                _capi.check(ctx.lib.edmp_sampler_set_allreduce(ctx.h, None, None))
                _read_stats()
            return res
        if isinstance(noise, str):
            if noise != "device":
                raise ValueError("noise must be an array, a device tensor, None (NumPy stream) or 'device'")
            _capi.check(
                ctx.lib.edmp_denoise_guided_rng_dev(ctx.h, int(seed) & 0xFFFFFFFFFFFFFFFF, batch_size, _capi.as_pd(s), _capi.as_pd(g),
                                                    1 if guide is not None else 0, int(t_stop), 1 if zero_row0 else 0, ptr(out)),
                "edmp_denoise_guided_rng_dev",
            )
            return ctx.hand_over(out) if return_device else ctx.to_host(out)
        if noise is None:
            # Reference contract: z comes from the GLOBAL NumPy RandomState, X_T first, then one draw per step
            # (diffusion.py:303, 126).  The stream is drawn in chunks of `chunk_steps` steps and each chunk is uploaded and
            # enqueued at once, so the host RNG (edmp_amd.nprng: ~0.9 ms per step for 1024 rows on 16 cores; NumPy itself needs 3.3-4 ms) runs while the GPU
            # denoises the previous chunk.  The numbers and their order are those of one big standard_normal call.
            guided = 1 if guide is not None else 0
            # chunk plan: (steps, carries X_T).  The first chunks are short and double (1, 2, 4, ... steps) so that the GPU
            # starts after ONE step's worth of draws and the host gets ahead of it geometrically.
            plan, t_left, first, k = [], self.T - int(t_stop), True, 1
            while t_left > 0:
                kk = min(k, int(chunk_steps), t_left)
                plan.append((kk, first))
                t_left -= kk
                first = False
                k *= 2
            per_step = batch_size * num_channels * traj_len
            # The draws go straight into PINNED host memory (a ring of staging buffers owned by the context), so the upload is
            # one asynchronous DMA on the copy stream - no pageable-memory staging copy on a host core, which the draw threads
            # need.  A staging buffer is reused only after the copy out of it has completed (event).  The draws run in ONE
            # background thread (the C helper releases the GIL), strictly in order, with two threads fewer than the CPU quota
            # (nprng.draw_threads): chunk i+1 is drawn while this thread uploads chunk i and enqueues its kernel launches.
            ring = ctx.pinned_ring(3, (int(chunk_steps) + 1) * per_step)
            nthr = nprng.draw_threads()

            import os as _os
            import time as _time

            trace = self.noise_trace = [] if _os.environ.get("EDMP_NOISE_TRACE") else None  # per chunk: host timestamps (debug)
            t_call = _time.perf_counter()

            def draw(i):
                kk, f = plan[i]
                slot = ring[i % len(ring)]
                t0 = _time.perf_counter()
                if slot["event"] is not None:
                    slot["event"].synchronize()  # the previous upload out of this buffer is done
                t1 = _time.perf_counter()
                n = (kk + (1 if f else 0)) * per_step
                nprng.standard_normal((n,), nthreads=nthr, out=slot["np"][:n])
                if trace is not None:
                    trace.append(("draw", i, kk, t0 - t_call, t1 - t_call, _time.perf_counter() - t_call))
                return slot, n

            t_hi, keep = self.T, []
            pool = ctx.draw_pool()  # ONE long-lived draw thread per context: its OpenMP team stays alive (and warm) between scenes
            pending = pool.submit(draw, 0) if plan else None
            try:
                for i, (kk, f) in enumerate(plan):
                    ta = _time.perf_counter()
                    slot, n = pending.result()
                    tb = _time.perf_counter()
                    pending = pool.submit(draw, i + 1) if i + 1 < len(plan) else None
                    zd = ctx.upload_pinned(slot, n)  # copy stream; this context's stream waits for it
                    keep.append(zd)  # stays allocated until the stream has consumed it
                    last = i + 1 == len(plan)
                    tc = _time.perf_counter()
                    _capi.check(
                        ctx.lib.edmp_denoise_guided_segment_dev(ctx.h, ptr(zd), batch_size, _capi.as_pd(s), _capi.as_pd(g), guided, t_hi, t_hi - kk,
                                                                1 if f else 0, 1 if zero_row0 else 0, ptr(out) if last else None),
                        "edmp_denoise_guided_segment_dev",
                    )
                    if trace is not None:
                        ev = torch.cuda.Event(enable_timing=True)
                        ev.record(ctx.stream)
                        trace.append(("main", i, kk, ta - t_call, tb - t_call, tc - t_call, _time.perf_counter() - t_call, ev))
                    t_hi -= kk
            except BaseException:
                # leave nothing in flight that still reads the staging ring or the chunk tensors: the draw thread finishes its
                # current chunk, the stream drains, then the error propagates
                if pending is not None:
                    try:
                        pending.result()
                    except Exception:
                        pass
                try:
                    ctx.sync()
                except Exception:
                    pass
                raise
            if return_device:
                ctx.sync()
                return out
            res = ctx.to_host(out)
            del keep
            return res
        stream = noise if isinstance(noise, PinnedNoiseStream) else None
        if stream is not None:
            noise = stream.tensor
        if isinstance(noise, torch.Tensor) and not noise.is_cuda and noise.is_pinned() and noise.dtype == torch.float64 and noise.is_contiguous():
            # A pre-drawn (or, PinnedNoiseStream, still being drawn) stream in PAGE-LOCKED host memory (infer_serial's scene-ahead feeder): uploaded in the same doubling chunks as the
            # on-the-fly path (1, 2, 4, ... chunk_steps steps; X_T rides with the first), every copy queued at once on the copy stream and each
            # segment of the loop ordered after its chunk - the GPU starts after 5.7 MB instead of after the whole 734 MB, and no host core
            # draws or copies anything while the loop runs.
            if tuple(noise.shape) != (self.T + 1, batch_size, num_channels, traj_len):
                raise ValueError(f"noise must be f64 {(self.T + 1, batch_size, num_channels, traj_len)}, got {tuple(noise.shape)}")
            guided = 1 if guide is not None else 0
            flat, per_step = noise.view(-1), batch_size * num_channels * traj_len
            t_hi, off, k, first, keep = self.T, 0, 1, True, []
            while t_hi > int(t_stop):
                kk = min(k, int(chunk_steps), t_hi - int(t_stop))
                n = (kk + (1 if first else 0)) * per_step
                if stream is not None:
                    stream.wait_until(off + n)  # (only the first scene of a run ever waits here: the feeder works a whole scene ahead)
                zd = ctx.upload_pinned({"t": flat[off:off + n]}, n)
                keep.append(zd)
                last = t_hi - kk == int(t_stop)
                _capi.check(
                    ctx.lib.edmp_denoise_guided_segment_dev(ctx.h, ptr(zd), batch_size, _capi.as_pd(s), _capi.as_pd(g), guided, t_hi, t_hi - kk,
                                                            1 if first else 0, 1 if zero_row0 else 0, ptr(out) if last else None),
                    "edmp_denoise_guided_segment_dev",
                )
                t_hi, off, k, first = t_hi - kk, off + n, k * 2, False
            if return_device:
                ctx.sync()  # (the chunk tensors die with this frame)
                return out
            res = ctx.to_host(out)
            del keep
            return res
        nd = ctx.adopt(noise) if (isinstance(noise, torch.Tensor) and noise.is_cuda) else ctx.to_dev(noise, torch.float64)
        if tuple(nd.shape) != (self.T + 1, batch_size, num_channels, traj_len) or nd.dtype != torch.float64:
            raise ValueError(f"noise must be f64 {(self.T + 1, batch_size, num_channels, traj_len)}, got {tuple(nd.shape)} {nd.dtype}")
        _capi.check(
            ctx.lib.edmp_denoise_guided_dev(ctx.h, ptr(nd), batch_size, _capi.as_pd(s), _capi.as_pd(g), 1 if guide is not None else 0, int(t_stop),
                                            1 if zero_row0 else 0, ptr(out)),
            "edmp_denoise_guided_dev",
        )
        if return_device:
            return ctx.hand_over(out)
        return ctx.to_host(out)

    def denoise(self, model, traj_len, num_channels, start=None, goal=None, condition=True, *, batch_size=1, noise=None):
        """diffusion.py:253-278 (unguided), batched; returns X[0] like the reference when batch_size == 1."""
        X = self.denoise_guided(model, None, traj_len, num_channels, None, batch_size=batch_size, start=start, goal=goal, condition=condition, noise=noise)
        return X[0] if batch_size == 1 else X

    def denoise_step(self, model, guide, X, z, t, start, goal, guidance_schedule=None, zero_row0=True, allreduce=None):
        """One teacher-forced reverse step on host arrays: returns dict(eps, x_post, grad (mixed, or None), x_out).
        ``allreduce(tensor)``: optional in-place sum over ranks of the device scalar sum(g^2) (multi-GPU)."""
        ctx = self.ctx
        B, Cc, N = X.shape
        self._prepare(model, guide, B, guidance_schedule)
        _capi.check(ctx.lib.edmp_sampler_set_condition(ctx.h, 1))
        Xd = ctx.to_dev(np.array(X, dtype=np.float64), torch.float64)  # a fresh device tensor: updated in place below
        zd = ctx.to_dev(np.asarray(z, dtype=np.float64), torch.float64)
        s, g = _startgoal(start, goal, needed=True)
        eps = ctx.empty((B, Cc, N), torch.float32)
        xpost = ctx.empty((B, Cc, N), torch.float64)
        grad = ctx.empty((B, Cc, N - 2), torch.float64)
        _capi.check(ctx.lib.edmp_step_a_dev(ctx.h, ptr(Xd), ptr(zd), B, int(t), _capi.as_pd(s), _capi.as_pd(g), 1 if zero_row0 else 0, ptr(eps), ptr(xpost)), "edmp_step_a_dev")
        guided = (t % 2) < 1 and t >= 5
        if guided and allreduce is not None:
            with torch.cuda.stream(ctx.stream):
                allreduce(self.sumsq_tensor())
        _capi.check(ctx.lib.edmp_step_b_dev(ctx.h, ptr(Xd), B, int(t), _capi.as_pd(s), _capi.as_pd(g), ptr(grad)), "edmp_step_b_dev")
        return dict(eps=ctx.to_host(eps), x_post=ctx.to_host(xpost), grad=ctx.to_host(grad) if guided else None, x_out=ctx.to_host(Xd))

    def device_noise(self, seed, step_index, batch_size, num_channels=7, traj_len=50) -> np.ndarray:
        """the (B,C,N) z tensor of ``noise="device"`` at step_index (0 = X_T, 1 + T - t = reverse step t)."""
        ctx = self.ctx
        out = ctx.empty((batch_size, num_channels, traj_len), torch.float64)
        _capi.check(ctx.lib.edmp_rng_normal_dev(ctx.h, int(seed) & 0xFFFFFFFFFFFFFFFF, int(step_index), batch_size, num_channels, traj_len, ptr(out)))
        return ctx.to_host(out)

    def sumsq_tensor(self) -> torch.Tensor:
        """zero-copy f64 view of the device scalar holding sum(g^2) of the last guided step."""
        p = self.ctx.lib.edmp_sumsq_ptr_dev(self.ctx.h)

        class _Holder:
            __cuda_array_interface__ = {"shape": (1,), "typestr": "<f8", "data": (int(p), False), "version": 2}

        return torch.as_tensor(_Holder(), device=self.ctx.device)
