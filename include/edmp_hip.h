/* edmp_hip.h — C-ABI of libedmp_hip.so: the MI355X (gfx950) implementation of EDMP's guided reverse-diffusion
 * sampler hot path.  This is the drop-in boundary: plain pointers and sizes, opaque context handle,
 * int status return (0 = ok, <0 = error; text via edmp_last_error()).  No torch / Python types.
 *
 * The reference (vishal-2000/EDMP) has no FFI: the path is reached through duck-typed Python objects created in
 * infer_serial.py:43-51,112 and passed to Diffusion.denoise_guided (infer_serial.py:134-143).  Each entry point
 * below names the reference method(s) it replaces (paths relative to the reference checkout); the Python binding
 * that exposes the reference's signatures on top of this ABI is edmp_amd/_capi.py (ctypes), see INTEGRATION.md.
 *
 * Conventions: pointers suffixed _dev are device (HBM) addresses on the context's GPU, all others are host
 * addresses read synchronously before the call returns.  All device work is stream-ordered on the context's
 * stream (edmp_ctx_set_stream); calls return without synchronising unless stated.  One host thread per context.
 * Trajectory tensors use the reference's layout (B, C=7, N) row-major, "f64" = IEEE double, "f32" = float.
 */
#ifndef EDMP_HIP_H
#define EDMP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EDMP_OK 0
#define EDMP_ERR_ARG (-1)
#define EDMP_ERR_HIP (-2)
#define EDMP_ERR_STATE (-3)
#define EDMP_ERR_LAYOUT (-4) /* edmp_unet_load_packed: the image was packed by another library version / under other builder switches */

#define EDMP_MAX_LEVELS 8
#define EDMP_N_JOINTS 7
#define EDMP_N_LINKS 9
#define EDMP_MAX_OBSTACLES 64

typedef struct edmp_ctx edmp_ctx;

/* ---- context ------------------------------------------------------------------------------------------- */
const char* edmp_last_error(void);
int edmp_version(void);
/* one context per GPU; creates its own stream */
int edmp_ctx_create(int device, edmp_ctx** out);
void edmp_ctx_destroy(edmp_ctx* ctx);
/* run on a caller-owned hipStream_t (e.g. torch's current stream); NULL restores the context's own stream */
int edmp_ctx_set_stream(edmp_ctx* ctx, void* hip_stream);
int edmp_ctx_synchronize(edmp_ctx* ctx);

/* ---- denoiser: TemporalUNet -------------------------------------------------------------------------- */
/* replaces TemporalUNet.__init__/load (diffusion/models/temporalunet.py:11-45, 88-92) */
typedef struct edmp_unet_desc {
    int32_t input_dim;             /* 7 */
    int32_t time_dim;              /* 32 */
    int32_t n_levels;              /* len(dims), 6 */
    int32_t dims[EDMP_MAX_LEVELS]; /* (32,64,128,256,512,512) */
    int32_t horizon;               /* N = 50 */
    int32_t T;                     /* number of diffusion steps the time-bias table covers (255) */
} edmp_unet_desc;

/* number of floats of the flat parameter blob for `desc` (all state-dict tensors, in state-dict order, each in
 * its native torch layout, concatenated) */
int64_t edmp_unet_param_count(const edmp_unet_desc* desc);
/* upload + repack weights, precompute the (T x sum Cout) time-bias table, allocate activations for max_batch */
int edmp_unet_load(edmp_ctx* ctx, const edmp_unet_desc* desc, const float* params, int64_t n_params, int max_batch);
/* Packed weight image = the device-side layout edmp_unet_load produces from the state dict (conv weights as MFMA
 * fragment streams / [tap][Cout][Cin], biases, GroupNorm affine, time-MLP weights), as ONE float blob that loads with a
 * single host-to-device copy (e.g. straight from an mmap'ed file) instead of re-reading, flattening and repacking 120 MB
 * (the reference's load is torch.load + load_state_dict, temporalunet.py:88-92).  edmp_unet_packed_size returns the
 * image's float count of the current model and the library's layout id; an image only loads into the same
 * architecture (desc) and layout id. */
int64_t edmp_unet_packed_size(edmp_ctx* ctx, int* layout);
int edmp_unet_read_packed(edmp_ctx* ctx, float* out_host, int64_t capacity);
int edmp_unet_load_packed(edmp_ctx* ctx, const edmp_unet_desc* desc, const float* packed, int64_t n_packed, int layout,
                          int max_batch);
/* replaces TemporalUNet.forward (temporalunet.py:47-76): x (B,C,N) f32, integer t in [1,T] -> eps (B,C,N) f32 */
int edmp_unet_forward_dev(edmp_ctx* ctx, const float* x_dev, int B, int t, float* eps_dev);
/* debug/parity: copy an internal activation, converted to the reference layout (B, C, L) f32.
 * which: 0..n_levels-1 = output of down level i, 100 = middle block, 200+i = output of up level i */
int edmp_unet_read_activation_dev(edmp_ctx* ctx, int which, int B, float* out_dev, int* C_out, int* L_out);
/* algorithmic FLOPs of one forward per trajectory: (a) nominal = every conv tap counted, the reference's own
 * arithmetic (SURVEY.md §8d, 187 339 904 for the full net); (b) executed = taps that fall in the zero padding are
 * skipped by the kernels */
int edmp_unet_flops(edmp_ctx* ctx, double* nominal, double* executed);
/* FLOPs per trajectory of the DIRECT convolution with the taps that only ever meet zero padding removed (round 1's
 * "executed" figure, 122.0 MFLOP for the full-size net).  edmp_unet_flops' `executed` counts the MFMA work actually
 * issued, which is lower where the L = 2 / L = 4 convolutions run in Karatsuba form (wide.hip). */
int edmp_unet_flops_direct(edmp_ctx* ctx, double* direct);
/* Where the issued MFMA work of one forward runs (no reference counterpart; bench.py's roofline): FLOPs per trajectory issued on
 * the fp32 matrix pipe (edmp_unet_flops' `executed` minus the work of the layers that run in bf16x3 form) and on the bf16 matrix
 * pipe (csrc/bf3.hip: six exact bf16 partial products per fp32 product, fp32 accumulation; 0 with EDMP_BF16X3=0). */
int edmp_unet_flops_pipes(edmp_ctx* ctx, double* f32_issued, double* bf16_issued);

/* ---- guide: IntersectionVolumeGuide -------------------------------------------------------------------- */
/* replaces IntersectionVolumeGuide.__init__/define_link_information/define_obstacles
 * (lib/guide.py:13-43, 243-342, 118-158).  obstacle_config (no,10) f64 rows [xyz, quat xyzw, full extents];
 * clearance/expansion (G,T) f64 = one row per distinct guide class; builds the device table of inflated obstacle
 * AABBs for every (class, t in 0..T).  link_half_extents (9,3) f32; dh (7,4) f32 rows [a,d,cos(alpha),sin(alpha)];
 * static_frames (9,3,4) f32. */
int edmp_scene_set(edmp_ctx* ctx, const double* obstacle_config, int n_obstacles, const double* clearance,
                   const double* expansion, int n_classes, int T, const float* link_half_extents, const float* dh,
                   const float* static_frames);
/* per-row parameters (infer_serial.py:56-91): class index, method (0 iv / 1 sv), grad_norm (0/1),
 * guidance_schedule (B,T) f64 */
int edmp_rows_set(edmp_ctx* ctx, const int32_t* row_class, const float* method, const double* grad_norm,
                  const double* guidance_schedule, int B, int T);
/* copy the obstacle AABB table entry (class, t): out (no, 6) f32 = [min xyz, max xyz] (parity checks) */
int edmp_scene_read_aabbs(edmp_ctx* ctx, int cls, int t, float* out_host);
/* replaces IntersectionVolumeGuide.cost (lib/guide.py:354-395): joints (n,7,L) f32 -> volumes (n,L,9*no) f32.
 * row r uses class row_class[r] if use_row_class else class 0; t = 0 means no inflation. */
int edmp_guide_cost_dev(edmp_ctx* ctx, const float* joints_dev, int n, int L, int t, int use_row_class,
                        float* volumes_dev);
/* replaces swept_volume_cost (lib/guide.py:473-537): joints (n,7,L) f32 interior waypoints, start/goal (7,) f32
 * -> volumes (n, L+1, 9*no) f32 */
int edmp_guide_swept_cost_dev(edmp_ctx* ctx, const float* joints_dev, int n, int L, int t, int use_row_class,
                              const float* start, const float* goal, float* volumes_dev);
/* replaces get_gradient (lib/guide.py:597-635) incl. the whole-batch norm mixing: joints (B,7,L) f64 (already
 * clipped) -> gradient (B,7,L) f64.  sumsq_dev (optional, may be NULL) receives sum(g^2) (f64) before mixing. */
int edmp_guide_gradient_dev(edmp_ctx* ctx, const double* joints_dev, int B, int L, const double* start,
                            const double* goal, int t, double* grad_dev, double* sumsq_dev);
/* replaces the volume part of choose_best_trajectory (lib/guide.py:637-653): X (B,7,N) f64 -> per-row t=0 swept
 * volume (B,) f32; the argmin (first on ties) is written to *best_index (host, synchronises) if not NULL */
int edmp_row_swept_volumes_dev(edmp_ctx* ctx, const double* X_dev, int B, int N, const double* start,
                               const double* goal, float* volumes_dev, int* best_index);
/* torch.argmin over n f32 values on the device, as choose_best_trajectory uses it (lib/guide.py:650): first index of the
 * minimum; NaN counts as the smallest value (the first NaN wins).  The selection step of edmp_row_swept_volumes_dev. */
int edmp_argmin_dev(edmp_ctx* ctx, const float* v_dev, int n, int* index_host);

/* ---- plan success: the reference's simulator check, restated geometrically ------------------------------------ */
/* The guide sees every obstacle as a box (cylinders enter as (r, r, h) boxes, datasets/load_test_dataset.py:136-139) but the
 * reference's success check spawns TRUE cylinders (RobotEnvironment.spawn_collision_cylinders, lib/environment.py:249-268:
 * radius = config[7], height = config[8], axis = local z).  kind[i] = 0 cuboid (default after edmp_scene_set) / 1 cylinder
 * for obstacle row i of the obstacle_config given to edmp_scene_set.  Only edmp_success_rows_dev reads it. */
int edmp_scene_set_shapes(edmp_ctx* ctx, const int32_t* kind, int n_obstacles);
/* stands for RobotEnvironment.benchmark_trajectory + check_collisions (lib/environment.py:632-680, 591-608), the success
 * tally of infer_serial.py:94-99,165-168, for EVERY row of a batch: X (B,7,N) f64 on the device; a row succeeds iff all its
 * waypoints lie inside the joint limits (:659-661) and none of the 9 link boxes (lib/guide.py:243-342, f64 modified-DH
 * poses) meets an obstacle - exact oriented-box test against cuboids, exact box / finite-cylinder test against cylinders -
 * at any waypoint or at any of `substeps` joint-space interpolated configurations per segment ((1 - s/S) q_i + (s/S) q_i+1,
 * s = 0..S-1; the last waypoint once).  pybullet itself is third-party and absent: a geometric stand-in, not the simulator.
 * dh_f64 (host, optional): (7,4) f64 rows [a, d, cos(alpha), sin(alpha)]; NULL widens the f32 table of edmp_scene_set.
 * Outputs (device, each optional): ok (B,) int32 0/1, first (B,) int32 = first colliding waypoint or -1, within (B,) int32
 * 0/1.  counts_host (optional, synchronises): [rows ok, rows within limits, rows collision-free, B]. */
int edmp_success_rows_dev(edmp_ctx* ctx, const double* X_dev, int B, int N, int substeps, const double* dh_f64, int32_t* ok_dev,
                          int32_t* first_dev, int32_t* within_dev, int32_t* counts_host);

/* ---- sampler: Diffusion ------------------------------------------------------------------------------ */
/* replaces Diffusion.__init__/schedule_variance (diffusion/diffusion.py:10-20, 37-49) */
int edmp_sampler_init(edmp_ctx* ctx, int T, double variance_thresh);
int edmp_sampler_read_schedule(edmp_ctx* ctx, double* beta, double* alpha, double* alpha_bar);
/* the `condition` argument of denoise_guided / denoise (diffusion.py:305-307, 347-349): pin the first / last waypoint to
 * start / goal (default on; the reference driver always passes True) */
int edmp_sampler_set_condition(edmp_ctx* ctx, int on);
/* replaces p_sample_using_posterior (diffusion.py:116-135) with the noise draw z made explicit.
 * X (B,C,N) f64 updated in place.  zero_row0: apply quirk Q3 (row 0 of z zeroed when t == 1). */
int edmp_psample_dev(edmp_ctx* ctx, double* X_dev, const float* eps_dev, const double* z_dev, int B, int C, int N,
                     int t, int zero_row0);
/* One reverse step of denoise_guided (diffusion.py:314-349) split at the only cross-row coupling:
 *   step_a: eps = UNet(f32(X), t); X <- posterior(X, eps, z); if guided(t): g = raw gradient(clip(X[:, :, 1:-1])),
 *           partial sum(g^2) -> device scalar.
 *   step_b: if guided(t): X[:, :, 1:-1] -= sched[:, t-1] * mix(g, ||g||); X[:, :, 0] = start; X[:, :, -1] = goal.
 * Between the two a multi-GPU caller may all-reduce edmp_sumsq_ptr_dev() (one f64).  Optional outputs (NULL to
 * skip): eps_out_dev (B,C,N) f32, xpost_out_dev (B,C,N) f64, grad_out_dev (B,C,N-2) f64 (mixed gradient). */
int edmp_step_a_dev(edmp_ctx* ctx, double* X_dev, const double* z_dev, int B, int t, const double* start,
                    const double* goal, int zero_row0, float* eps_out_dev, double* xpost_out_dev);
int edmp_step_b_dev(edmp_ctx* ctx, double* X_dev, int B, int t, const double* start, const double* goal,
                    double* grad_out_dev);
double* edmp_sumsq_ptr_dev(edmp_ctx* ctx);
/* replaces Diffusion.denoise_guided (diffusion.py:300-356): noise (T+1,B,C,N) f64 on device, noise[0] = initial
 * draw, noise[1 + T - t] = draw of step t.  X_out (B,C,N) f64.  guided = 0 runs the unguided loop
 * (Diffusion.denoise, diffusion.py:253-278, batched).  t_stop: run steps T..t_stop+1 (0 = all).
 * zero_row0: this shard holds global row 0 (quirk Q3). */
int edmp_denoise_guided_dev(edmp_ctx* ctx, const double* noise_dev, int B, const double* start, const double* goal,
                            int guided, int t_stop, int zero_row0, double* X_out_dev);

/* The same loop in segments (steps t_hi .. t_lo+1), for callers that produce the noise stream while the GPU works:
 * init != 0 starts a run at t_hi = T (noise_dev[0] is the X_T draw, then one (B,C,N) draw per step); init == 0 continues
 * from the state kept in the context (noise_dev[0] is the draw of step t_hi) and performs no host synchronisation.
 * X_out_dev may be NULL except for the last segment.  Used by Diffusion.denoise_guided to overlap NumPy's RandomState
 * (the reference's noise contract, ~0.85 s per 1024-row scene on the host) with the denoising itself. */
int edmp_denoise_guided_segment_dev(edmp_ctx* ctx, const double* noise_dev, int B, const double* start, const double* goal,
                                    int guided, int t_hi, int t_lo, int init, int zero_row0, double* X_out_dev);

/* Device noise source — explicitly NOT the reference's NumPy RandomState stream (that contract is served by
 * edmp_denoise_guided_dev): Philox4x32-10 counter RNG + Box-Muller inside the sampler kernels, no noise tensor, no
 * host draw, no upload.  Same loop otherwise (replaces diffusion.py:300-356 with z ~ N(0, I) drawn on the GPU).
 * edmp_rng_normal_dev materialises the z tensor (B,C,N) f64 the loop uses at step_index (0 = initial X_T,
 * 1 + T - t = reverse step t) so the two entry points can be cross-checked. */
int edmp_denoise_guided_rng_dev(edmp_ctx* ctx, uint64_t seed, int B, const double* start, const double* goal, int guided,
                                int t_stop, int zero_row0, double* X_out_dev);
int edmp_rng_normal_dev(edmp_ctx* ctx, uint64_t seed, int step_index, int B, int C, int N, double* out_dev);

/* Replay mode of the device-resident loop: on = 1 captures the stream work of one edmp_denoise_guided*_dev call
 * (255 reverse steps, ~16k kernel nodes) into a hipGraph the first time and replays it while the call's arguments,
 * scene, rows and weights stay the same (start/goal are read from a device buffer and may change freely).  Results are
 * bit-identical to the eager enqueue.  Off by default:
 * measured neutral on MI355X at B = 4..1024 - the loop is bound by kernel execution, not by launch. */
int edmp_sampler_set_graph(edmp_ctx* ctx, int on);

/* ---- training-side forward process --------------------------------------------------------- */
/* replaces the arithmetic of Diffusion.q_sample (diffusion/diffusion.py:52-77, cumulative = 0: a = alpha),
 * Diffusion.q_sample_from_x0 (:79-105, cumulative = 1: a = alpha_bar) and the conditioning of generate_q_sample
 * (:239-242):   xt = sqrt(a[t_b - 1]) * x + sqrt(1 - a[t_b - 1]) * eps,   mean = sqrt(a[t_b - 1]) * x
 * with one timestep per row (t_host: B int32 on the HOST, each in 1..T); condition != 0 then pins xt[:, :, 0] and
 * xt[:, :, -1] to x.  x, eps, xt, mean: (B,C,N) f64 on the device; mean_dev may be NULL.  Rounded like NumPy's f64
 * expression (two products, one sum, no FMA contraction), so results are bit-identical to the reference's. */
int edmp_q_sample_dev(edmp_ctx* ctx, const double* x_dev, const double* eps_dev, const int32_t* t_host, int B, int C, int N,
                      int cumulative, int condition, double* xt_dev, double* mean_dev);

/* ---- resident objects ----------------------------------------------------------------------------------- */
/* The reference keeps Python objects alive side by side: one TemporalUNet per process, one IntersectionVolumeGuide per
 * scene (infer_serial.py:50, 112).  A context holds up to 3 models and 8 guides (scene tables + row arrays) resident
 * in HBM, addressed by a caller-chosen non-zero key; edmp_unet_load / edmp_scene_set / edmp_rows_set always act on the
 * CURRENT slot.  edmp_*_slot(key) makes `key` current (parking the previous one, evicting the least recently used
 * beyond the capacity) and returns 1 if the slot already holds a loaded object - nothing to upload -, 0 if it is
 * empty (load into it next), < 0 on error.  Key 0 is the default slot of callers that never use slots. */
int edmp_unet_slot(edmp_ctx* ctx, uint64_t key);
int edmp_guide_slot(edmp_ctx* ctx, uint64_t key);

/* ---- one logical batch over several GPUs ------------------------------------------------------------------ */
/* The reference has no distributed code; its only coupling between batch rows is the whole-batch gradient norm
 * gradient1 / np.linalg.norm(gradient1) (lib/guide.py:629).  When one reference batch is row-sharded over ranks, the
 * device-resident loop calls `fn(user, hip_stream, sumsq_dev)` once per guided step, between the gradient kernels
 * and the state update: the callee must enqueue, ON THAT STREAM, an in-place sum over ranks of the f64 device scalar
 * (e.g. ncclAllReduce / torch.distributed.all_reduce with that stream current) and return 0.  fn = NULL (default)
 * restores the single-GPU behaviour.  hipGraph replay is disabled while a caller hook is installed (an arbitrary callee is not
 * capturable; the native RCCL hook below is). */
typedef int (*edmp_allreduce_fn)(void* user, void* hip_stream, double* sumsq_dev);
int edmp_sampler_set_allreduce(edmp_ctx* ctx, edmp_allreduce_fn fn, void* user);

/* The hook as native code (csrc/rccl_hook.hip): one ncclAllReduce of the f64 scalar on the context's stream per guided step - no
 * Python / GIL inside the device-resident loop (the round-5 hook was a ctypes callback into torch.distributed.all_reduce:
 * edmp_amd/diffusion.py; it stays available through edmp_sampler_set_allreduce).  RCCL is resolved at run time, never linked:
 *   edmp_rccl_load(path)       path = NULL: the RCCL already in the process (the one torch brought), else librccl.so.1; or a path
 *   edmp_rccl_unique_id(id)    ncclGetUniqueId into 128 caller bytes: one rank calls it and hands the bytes to the others by any
 *                              means (torch.distributed.broadcast_object_list over gloo or nccl, MPI, a file)
 *   edmp_rccl_attach(ctx, id, nranks, rank)   ncclCommInitRank on the context's device (collective over the ranks), installs the hook
 *   edmp_rccl_attach_comm(ctx, comm)          borrow a communicator the host already has (torch: ProcessGroupNCCL._comm_ptr())
 *   edmp_rccl_enable(ctx, on)  attach leaves the hook ON; 0 switches the collective off and keeps the communicator (runs of this
 *                              context that are NOT shards of one logical batch), 1 switches it on again
 *   edmp_rccl_detach(ctx)      remove the hook, destroy an owned communicator (also done by edmp_ctx_destroy).  edmp_sampler_set_allreduce
 *                              replaces the ACTIVE hook only; an attached communicator stays and edmp_rccl_enable brings it back
 *   edmp_rccl_info(ctx, out)   out = {nranks, rank, 0 none | 1 own communicator | 2 borrowed}
 * ncclAllReduce is stream-capturable: whole-run hipGraph replay (edmp_sampler_set_graph) stays legal with THIS hook installed.
 * edmp_sampler_allreduce_stats: host time spent inside the hook (any hook) since the last reset: {calls, total ns, max ns}. */
int edmp_rccl_load(const char* path);
int edmp_rccl_unique_id(void* id128);
int edmp_rccl_attach(edmp_ctx* ctx, const void* id128, int nranks, int rank);
int edmp_rccl_attach_comm(edmp_ctx* ctx, void* nccl_comm);
int edmp_rccl_enable(edmp_ctx* ctx, int on);
int edmp_rccl_detach(edmp_ctx* ctx);
int edmp_rccl_info(edmp_ctx* ctx, int32_t out[3]);
int edmp_sampler_allreduce_stats(edmp_ctx* ctx, uint64_t out[3], int reset);

/* ---- instrumentation ----------------------------------------------------------------------------------- */
/* accumulate HIP-event time of the dominant kernel family (the MFMA conv kernels of the UNet layer program) while enabled:
 * on = 1: one event pair around every conv launch (per-op table, edmp_prof_ops; ~2 events of overhead per launch);
 * on = 2: one event pair around the whole layer program of each reverse step (the family's total, negligible overhead);
 * on = 0: off.  Events are recorded on the context's stream. */
int edmp_prof_enable(edmp_ctx* ctx, int on);
/* total ms and launch count of the MFMA conv kernels since the last reset (synchronises) */
int edmp_prof_read(edmp_ctx* ctx, double* conv_ms, int64_t* conv_launches, int reset);
/* Per-op view of the same instrumentation (bench.py's per-kernel roofline table; no reference counterpart): for every
 * op i < min(*n_ops, cap) of the loaded UNet's layer program: summed event time [ms], launches, executed FLOPs per
 * trajectory per launch, and the kernel instance name (64 bytes each, as rocprofv3 prints it without "edmp::"). */
int edmp_prof_ops(edmp_ctx* ctx, int cap, int* n_ops, double* ms, int64_t* calls, double* flops_exec, char* names);
/* per op of the same program: FLOPs per trajectory per launch issued on the bf16 matrix pipe (0 for an fp32-MFMA op) */
int edmp_prof_ops_bf16(edmp_ctx* ctx, int cap, int* n_ops, double* flops_bf16);

#ifdef __cplusplus
}
#endif
#endif /* EDMP_HIP_H */
