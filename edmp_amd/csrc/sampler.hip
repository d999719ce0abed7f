// sampler.hip — the guided reverse-diffusion loop (Diffusion.denoise_guided) and the C-ABI context plumbing.
//
// Replaces (reference diffusion/diffusion.py): __init__/schedule_variance :10-20,37-49, p_sample_using_posterior
// :116-135, clip_joints :280-298 (inside the guide kernel), denoise_guided :300-356, denoise :253-278.
// State X is float64 on device exactly like the reference's NumPy state; the UNet sees float32(X), eps (f32) is
// promoted to f64 in the posterior; the noise stream z is an input (host NumPy RNG order is part of the contract).
#include "common.h"
#include "tail.h"

namespace edmp {

static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

// unet.hip / guide.hip
int unet_forward_impl(edmp_ctx* ctx, const float* x_dev, int B, int t, float* eps_dev);
int unet_run_program(edmp_ctx* ctx, int B, int t, const TailP* tail, bool* tail_done);
int guide_raw_gradient_from_X(edmp_ctx* ctx, const double* X_dev, int B, int N, int t, bool reduce);
const double* guide_rowsq(edmp_ctx* ctx);
int guide_set_startgoal(edmp_ctx* ctx, const double* start, const double* goal);
int guide_prepare(edmp_ctx* ctx, int B, int L);
const float* guide_graw(edmp_ctx* ctx);
const double* guide_grad_norm(edmp_ctx* ctx);
const double* guide_sched(edmp_ctx* ctx);
double* guide_sumsq(edmp_ctx* ctx);
int guide_rows_T(edmp_ctx* ctx);

struct Sampler {
    int T = 0;
    std::vector<double> beta, alpha, alpha_bar, c1, sqrt_alpha;  // host tables (f64)
    // scratch for the loop
    double* X = nullptr;    // (B,C,N) f64 loop state
    double* sg = nullptr;   // [14] start|goal f64
    int run_B = 0;          // batch of the run whose state sits in X (segmented runs)
    int condition = 1;      // pin X[:, :, 0] / X[:, :, -1] to start / goal (diffusion.py:305-307, 347-349)
    int cap = 0;            // elements
    // whole-run hipGraph (edmp_sampler_set_graph): the enqueue of one denoise_loop call captured once and replayed while the
    // call's arguments stay the same (start/goal travel through `sg`, so they are not part of the key)
    struct GraphKey {
        const void* noise = nullptr;
        const void* hook = nullptr;  // the (native) all-reduce hook's communicator record the capture contains, or null
        uint64_t seed = 0;
        int use_rng = 0, B = 0, guided = 0, t_hi = 0, t_lo = 0, init = 0, zero_row0 = 0, condition = 0;
        uint64_t epoch = 0;  // bumped by anything that invalidates captured pointers/arguments (scene, rows, weights)
        bool operator==(const GraphKey& o) const {
            return noise == o.noise && hook == o.hook && seed == o.seed && use_rng == o.use_rng && B == o.B && guided == o.guided && t_hi == o.t_hi &&
                   t_lo == o.t_lo && init == o.init && zero_row0 == o.zero_row0 && condition == o.condition && epoch == o.epoch;
        }
    } gkey;
    hipGraphExec_t gexec = nullptr;
    int graph_on = 0;   // edmp_sampler_set_graph
    int graph_captures = 0, graph_replays = 0;
    double* qcoef = nullptr;  // [B][2] sqrt(a), sqrt(1 - a) of edmp_q_sample_dev
    int qcoef_cap = 0;
    // "one logical batch over several GPUs": called between the two halves of every guided step of the device-resident
    // loop to sum the device scalar sum(g^2) over ranks (lib/guide.py:629 is the only coupling between rows)
    edmp_allreduce_fn ar_fn = nullptr;
    void* ar_user = nullptr;
    struct RcclHook* rccl = nullptr;              // rccl_hook.hip: the native hook's communicator, when that hook is installed
    uint64_t ar_calls = 0, ar_ns = 0, ar_max_ns = 0;  // host time spent inside the hook (edmp_sampler_allreduce_stats)
};

void sampler_rccl_destroy(Sampler* s);            // rccl_hook.hip
bool sampler_hook_is_native(const Sampler* s);    // rccl_hook.hip: the installed hook is the stream-capturable ncclAllReduce one

void sampler_destroy(Sampler* s) {
    if (!s) return;
    sampler_rccl_destroy(s);
    for (void* p : {(void*)s->X, (void*)s->sg, (void*)s->qcoef})
        if (p) (void)hipFree(p);
    if (s->gexec) (void)hipGraphExecDestroy(s->gexec);
    delete s;
}

// x <- (x - c1 * eps) / sqrt(alpha) + beta * z ; quirk Q3: z of (global) row 0 is zeroed at t == 1
__global__ void psample_kernel(double* __restrict__ X, const float* __restrict__ eps, const double* __restrict__ z, int n, int per_row,
                               double c1, double sqrt_alpha, double beta, int zero_row0) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double zz = z[i];
    if (zero_row0 && i < per_row) zz = 0.0;
    X[i] = (X[i] - c1 * (double)eps[i]) / sqrt_alpha + beta * zz;
}

// X[:, :, 1:-1] -= sched[:, t-1] * ((1-gn) g + gn g/||g||);  then X[:, :, 0] = start, X[:, :, -1] = goal
// rowsq != nullptr: the whole-batch sum(g^2) is formed here from the per-row partials, by every block in the summation order
// of reduce_rowsq_kernel (bit-identical) - one launch less per guided step of the device-resident loop
__global__ __launch_bounds__(256) void update_kernel(double* __restrict__ X, const float* __restrict__ graw, const double* __restrict__ sumsq,
                              const double* __restrict__ rowsq, const double* __restrict__ grad_norm, const double* __restrict__ sched, int sched_T, int t, int B, int C, int N,
                              const double* __restrict__ sg, int guided, double* __restrict__ grad_out, float* __restrict__ xin, int cond, int n) {
    __shared__ double sm[256];
    double total = 0.0;
    if (guided) total = rowsq ? block_sum_rowsq(rowsq, B, sm) : sumsq[0];
    // (one element per thread: a grid-stride variant with 256 blocks - the redundant reduction paid 256 instead of 1344 times - was
    // measured SLOWER, 11.6 vs 7.9 us: five dependent f64 round trips per thread at one wave per SIMD; round 5)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int l = i % N;
    const int c = (i / N) % C;
    const int b = i / (N * C);
    double xnew = X[i];
    if (l == 0) {
        if (cond) {
            xnew = sg[c];
            X[i] = xnew;
        }
    } else if (l == N - 1) {
        if (cond) {
            xnew = sg[7 + c];
            X[i] = xnew;
        }
    } else if (guided) {
        const int L = N - 2;
        const size_t gi = ((size_t)b * C + c) * L + (l - 1);
        const float nrm = (float)sqrt(total);
        const float gv = graw[gi];
        const double gn = grad_norm[b];
        const double mixed = (1.0 - gn) * (double)gv + gn * (double)(gv / nrm);
        xnew = xnew - sched[(size_t)b * sched_T + (t - 1)] * mixed;
        X[i] = xnew;
        if (grad_out) grad_out[gi] = mixed;
    }
    if (xin) {  // next step's UNet input [B][N][8] (channel 7 is the zero pad, written by the c == 0 thread)
        float* o = xin + ((size_t)b * N + l) * 8;
        o[c] = (float)xnew;
        if (c == 0) o[7] = 0.0f;
    }
}

__global__ void condition_kernel(double* __restrict__ X, int B, int C, int N, const double* __restrict__ sg) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;  // over B*C
    if (i >= B * C) return;
    int c = i % C;
    X[(size_t)i * N] = sg[c];
    X[(size_t)i * N + N - 1] = sg[7 + c];
}

// the z tensor (B,C,N) f64 the loop uses at `step` (0 = initial state, 1 + T - t = reverse step t): tests / inspection
__global__ void rng_normal_kernel(uint64_t seed, int step, double* __restrict__ out, int B, int C, int N) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * N) return;
    const int b = i / N, l = i - b * N;
    float z[8];
    rng_normal8(seed, (uint32_t)step, (uint32_t)i, z);
    for (int c = 0; c < C && c < 8; ++c) out[((size_t)b * C + c) * N + l] = (double)z[c];
}

// X_T = N(0, I) with start/goal conditioning, plus the first UNet input              (diffusion.py:303-307)
__global__ void init_state_rng_kernel(uint64_t seed, double* __restrict__ X, float* __restrict__ xin, const double* __restrict__ sg, int B, int C,
                                      int N, int cond) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * N) return;
    const int b = i / N, l = i - b * N;
    float z[8];
    rng_normal8(seed, 0u, (uint32_t)i, z);
    float xo[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C && c < 8; ++c) {
        double x = (double)z[c];
        if (cond && l == 0) x = sg[c];
        if (cond && l == N - 1) x = sg[7 + c];
        X[((size_t)b * C + c) * N + l] = x;
        xo[c] = (float)x;
    }
    float4* o = reinterpret_cast<float4*>(xin + (size_t)i * 8);
    o[0] = make_float4(xo[0], xo[1], xo[2], xo[3]);
    o[1] = make_float4(xo[4], xo[5], xo[6], xo[7]);
}

// X (B,C,N) f64 -> UNet input [B][N][8] f32 (channels >= C zero).  Thread per (b, l).
__global__ void pack_state_kernel(const double* __restrict__ X, float* __restrict__ xin, int B, int C, int N) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * N) return;
    const int b = i / N, l = i - b * N;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = (c < C) ? (float)X[((size_t)b * C + c) * N + l] : 0.0f;
    float4* o = reinterpret_cast<float4*>(xin + (size_t)i * 8);
    o[0] = make_float4(v[0], v[1], v[2], v[3]);
    o[1] = make_float4(v[4], v[5], v[6], v[7]);
}

// Tail of the denoiser + posterior step in one launch, thread per (sample, waypoint):
//   eps[c] = final 1x1 conv of the UNet (final_conv.1, temporalunet.py:36) on h[b][l][0..Cin)
//   X <- (X - c1 eps)/sqrt(alpha) + beta z                                   (diffusion.py:116-135)
//   FINISH (steps without guidance): X[:, :, 0] = start, X[:, :, -1] = goal  (diffusion.py:347-349) and the next
//   step's UNet input [B][N][8] f32 is written, so an unguided reverse step is exactly UNet + this kernel.
template <bool FINISH, bool RNG, int CIN>
__global__ __launch_bounds__(256) void head_psample_kernel(const float* __restrict__ h, const float* __restrict__ w, const float* __restrict__ bias,
                                                           double* __restrict__ X, const double* __restrict__ z, float* __restrict__ eps_out,
                                                           float* __restrict__ xin, const double* __restrict__ sg, int B, int N, int Cin, int C,
                                                           double c1, double sqrt_alpha, double beta, int zero_row0, uint64_t seed, int rng_step, int cond) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (CIN > 0) {  // compile-time width: all input loads are issued back to back; the shared tail does the rest
        // the head's weights through LDS (224 wave-uniform scalar loads in a row serialise on the scalar cache)
        __shared__ float sw[8 * (CIN > 0 ? CIN : 4) + 8];
        for (int k = threadIdx.x; k < C * CIN; k += blockDim.x) sw[k] = w[k];
        if (threadIdx.x < 8) sw[8 * CIN + threadIdx.x] = (int)threadIdx.x < C ? bias[threadIdx.x] : 0.0f;
        const bool mine = i < B * N;
        const int ii = mine ? i : 0;
        const int b = ii / N, l = ii - b * N;
        const float* hp = h + (size_t)ii * Cin;
        float4 hv[CIN / 4];
        double xv[8], zv[8];
#pragma unroll
        for (int q = 0; q < CIN / 4; ++q) hv[q] = *reinterpret_cast<const float4*>(hp + 4 * q);
        tail_fetch(X, z, RNG, b, l, N, C, xv, zv);
        __syncthreads();
        if (mine)
            head_psample_item<FINISH, RNG, (CIN > 0 ? CIN : 4)>(hv, xv, zv, i, b, l, sw, sw + 8 * CIN, X, eps_out, xin, sg, N, C, c1, sqrt_alpha, beta, zero_row0, seed, rng_step, cond);
        return;
    }
    if (i >= B * N) return;
    const int b = i / N, l = i - b * N;
    const float* hp = h + (size_t)i * Cin;
    float xo[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float zr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (RNG) rng_normal8(seed, (uint32_t)rng_step, (uint32_t)i, zr);
    // each thread's Cin inputs are read ONCE (float4) and reused by all C outputs; weights are wave-uniform (scalar loads)
    float acc[8];
#pragma unroll
    for (int co = 0; co < 8; ++co) acc[co] = (co < C) ? bias[co] : 0.0f;
    {  // run-time width (architectures whose head input is neither 16 nor 32 channels wide)
        for (int c4 = 0; c4 < Cin; c4 += 4) {
            const float4 hv = *reinterpret_cast<const float4*>(hp + c4);
#pragma unroll
            for (int co = 0; co < 8; ++co) {
                if (co < C) {
                    const float* wr = w + co * Cin + c4;
                    acc[co] = fmaf(hv.x, wr[0], acc[co]);
                    acc[co] = fmaf(hv.y, wr[1], acc[co]);
                    acc[co] = fmaf(hv.z, wr[2], acc[co]);
                    acc[co] = fmaf(hv.w, wr[3], acc[co]);
                }
            }
        }
    }
#pragma unroll
    for (int co = 0; co < 8; ++co) {
        if (co >= C) break;
        const float a = acc[co];
        const size_t idx = ((size_t)b * C + co) * N + l;
        if (eps_out) eps_out[idx] = a;
        double zz = RNG ? (double)zr[co] : z[idx];
        if (zero_row0 && b == 0) zz = 0.0;
        double x = (X[idx] - c1 * (double)a) / sqrt_alpha + beta * zz;
        if (FINISH) {
            if (cond && l == 0) x = sg[co];
            if (cond && l == N - 1) x = sg[7 + co];
            xo[co] = (float)x;
        }
        X[idx] = x;
    }
    if (FINISH) {
        float4* o = reinterpret_cast<float4*>(xin + (size_t)i * 8);
        o[0] = make_float4(xo[0], xo[1], xo[2], xo[3]);
        o[1] = make_float4(xo[4], xo[5], xo[6], xo[7]);
    }
}

// forward process: xt = sa*x + sb*eps per row (coef = [B][2]); products and sum rounded separately (NumPy's evaluation)
__global__ void q_sample_kernel(const double* __restrict__ x, const double* __restrict__ eps, const double* __restrict__ coef,
                                double* __restrict__ xt, double* __restrict__ mean, int n, int per_row, int N, int condition) {
#pragma clang fp contract(off)  // hipcc fuses a*b + c into an FMA by default (and __dmul_rn / __dadd_rn are plain * and +)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = i / per_row, w = i % N;
    const double xv = x[i];
    const double m = coef[2 * b] * xv;
    const double e = coef[2 * b + 1] * eps[i];
    double v = m + e;
    if (condition && (w == 0 || w == N - 1)) v = xv;
    xt[i] = v;
    if (mean) mean[i] = m;
}

static int ensure_sampler_scratch(edmp_ctx* ctx, int n) {
    Sampler* s = ctx->sampler;
    if (s->cap >= n) return EDMP_OK;
    ctx->epoch++;
    if (s->X) (void)hipFree(s->X);
    s->X = nullptr;
    EDMP_HIP_CHECK(hipMalloc((void**)&s->X, (size_t)n * sizeof(double)));
    s->cap = n;
    return EDMP_OK;
}

static bool guided_step(int t) { return (t % 2) < 1 && t >= 5; }  // diffusion.py:311,326-327 (period 2)

static int set_startgoal(edmp_ctx* ctx, const double* start, const double* goal, bool need_guide) {
    Sampler* s = ctx->sampler;
    double sg[14];
    for (int i = 0; i < 7; ++i) {
        sg[i] = start[i];
        sg[7 + i] = goal[i];
    }
    EDMP_HIP_CHECK(hipMemcpyAsync(s->sg, sg, sizeof(sg), hipMemcpyHostToDevice, ctx->stream));
    EDMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (need_guide) return guide_set_startgoal(ctx, start, goal);
    return EDMP_OK;
}

// One reverse step, first half.  `fused` (the device-resident loop): the UNet input of this step already sits in
// unet->x_in (written by the previous step's tail kernel) and, on steps without guidance, the tail kernel also applies
// the start/goal conditioning and writes the next input.  Otherwise (teacher-forced API): X comes from the caller, is
// packed here, and conditioning is left to step_b so that the un-conditioned posterior can be returned.
static int step_a(edmp_ctx* ctx, double* X, const double* z, int B, int t, int zero_row0, int guided, float* eps_out, double* xpost_out,
                  bool fused, bool use_rng = false, uint64_t seed = 0) {
    Sampler* s = ctx->sampler;
    UNet* u = ctx->unet;
    const int C = u->desc.input_dim, N = u->desc.horizon;
    hipStream_t st = ctx->stream;
    float* xin = u->x_in;
    const float* hlast = u->h_last;
    const int n = B * C * N;
    const dim3 grid_bn((B * N + 255) / 256);
    if (!fused) hipLaunchKernelGGL(pack_state_kernel, grid_bn, dim3(256), 0, st, X, xin, B, C, N);
    const bool g = guided && guided_step(t);
    const int zr = (zero_row0 && t == 1) ? 1 : 0;  // Q3: row 0 only
    const int rstep = 1 + (s->T - t);
    // device-resident loop: the step's tail (final 1x1 conv, posterior, conditioning, next input) rides in the UNet's last
    // launch when the architecture ends in the fused final level (tail.h); the teacher-forced API keeps the separate launch
    TailP tail;
    bool tail_done = false;
    const bool want_tail = fused && !eps_out && !xpost_out;
    if (want_tail) {
        tail.X = X;
        tail.z = z;
        tail.xin = g ? nullptr : xin;
        tail.sg = s->sg;
        tail.C = C;
        tail.N = N;
        tail.c1 = s->c1[t - 1];
        tail.sqrt_alpha = s->sqrt_alpha[t - 1];
        tail.beta = s->beta[t - 1];
        tail.zero_row0 = zr;
        tail.seed = seed;
        tail.rng_step = rstep;
        tail.cond = s->condition;
        tail.finish = g ? 0 : 1;
        tail.rng = use_rng ? 1 : 0;
    }
    int rc = unet_run_program(ctx, B, t, want_tail ? &tail : nullptr, &tail_done);
    if (rc) return rc;
#define EDMP_HP_ARGS(xin_ptr) hlast, u->head_w, u->head_b, X, z, eps_out, (xin_ptr), s->sg, B, N, u->head_cin, C, s->c1[t - 1], s->sqrt_alpha[t - 1], s->beta[t - 1], zr, seed, rstep, s->condition
#define EDMP_HP_LAUNCH(FIN, RN, xin_ptr)                                                                                              \
    {                                                                                                                                  \
        if (u->head_cin == 32) hipLaunchKernelGGL((head_psample_kernel<FIN, RN, 32>), grid_bn, dim3(256), 0, st, EDMP_HP_ARGS(xin_ptr));      \
        else if (u->head_cin == 16) hipLaunchKernelGGL((head_psample_kernel<FIN, RN, 16>), grid_bn, dim3(256), 0, st, EDMP_HP_ARGS(xin_ptr)); \
        else hipLaunchKernelGGL((head_psample_kernel<FIN, RN, 0>), grid_bn, dim3(256), 0, st, EDMP_HP_ARGS(xin_ptr));                        \
    }
    if (tail_done) {
        // the UNet's last launch has already run the tail of this step
    } else if (fused && !g) {
        if (use_rng) EDMP_HP_LAUNCH(true, true, xin)
        else EDMP_HP_LAUNCH(true, false, xin)
    } else {
        if (use_rng) EDMP_HP_LAUNCH(false, true, (float*)nullptr)
        else EDMP_HP_LAUNCH(false, false, (float*)nullptr)
    }
#undef EDMP_HP_LAUNCH
#undef EDMP_HP_ARGS
    EDMP_HIP_CHECK(hipGetLastError());
    if (xpost_out) EDMP_HIP_CHECK(hipMemcpyAsync(xpost_out, X, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (g) {
        rc = guide_raw_gradient_from_X(ctx, X, B, N, t, /*reduce=*/!(fused && !s->ar_fn));
        if (rc) return rc;
    }
    return EDMP_OK;
}

static int step_b(edmp_ctx* ctx, double* X, int B, int t, int guided, double* grad_out, bool fused) {
    Sampler* s = ctx->sampler;
    UNet* u = ctx->unet;
    const int C = u->desc.input_dim, N = u->desc.horizon;
    hipStream_t st = ctx->stream;
    if (guided && guided_step(t)) {
        const int n = B * C * N;
        hipLaunchKernelGGL(update_kernel, dim3((n + 255) / 256), dim3(256), 0, st, X, guide_graw(ctx), guide_sumsq(ctx), (fused && !s->ar_fn) ? guide_rowsq(ctx) : nullptr, guide_grad_norm(ctx),
                           guide_sched(ctx), guide_rows_T(ctx), t, B, C, N, s->sg, 1, grad_out, fused ? u->x_in : nullptr, s->condition, n);
    } else if (!fused && s->condition) {
        hipLaunchKernelGGL(condition_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, X, B, C, N, s->sg);
    }  // fused + unguided: head_psample_kernel<true> already conditioned X and wrote the next input
    EDMP_HIP_CHECK(hipGetLastError());
    return EDMP_OK;
}

}  // namespace edmp

using namespace edmp;

extern "C" const char* edmp_last_error(void) { return g_err.c_str(); }
extern "C" int edmp_version(void) { return 100; }

extern "C" int edmp_ctx_create(int device, edmp_ctx** out) {
    EDMP_REQUIRE(out, "edmp_ctx_create: null out pointer");
    int ndev = 0;
    EDMP_HIP_CHECK(hipGetDeviceCount(&ndev));
    EDMP_REQUIRE(device >= 0 && device < ndev, "device %d not present (%d visible)", device, ndev);
    EDMP_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    EDMP_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; libedmp_hip is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return EDMP_ERR_STATE;
    }
    edmp_ctx* c = new edmp_ctx();
    c->device = device;
    EDMP_HIP_CHECK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    *out = c;
    return EDMP_OK;
}

extern "C" void edmp_ctx_destroy(edmp_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    unet_destroy(ctx->unet);
    guide_destroy(ctx, ctx->guide);
    for (auto& e : ctx->unet_slots) unet_destroy(e.second);
    for (auto& e : ctx->guide_slots) guide_destroy(ctx, e.second);
    sampler_destroy(ctx->sampler);
    ctx_pool_destroy(ctx);
    for (auto& e : ctx->prof.pending) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    for (auto& e : ctx->prof.pool) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

// make slot `key` the current one: park the current object (most recently used first), fetch the slot if resident
// Returns 1 only for a COMPLETE object: one whose load failed half way (edmp_scene_set after its `new Guide`, a UNet without a
// layer program) is destroyed and reported as an empty slot, so the caller loads into it again.
template <class T, class D, class C>
static int select_slot(std::vector<std::pair<uint64_t, T*>>& slots, T*& cur, uint64_t& cur_key, uint64_t key, int cap, D destroy, C complete) {
    auto usable = [&]() {
        if (cur && !complete(cur)) {
            destroy(cur);
            cur = nullptr;
        }
        return cur ? 1 : 0;
    };
    if (key == cur_key) return usable();
    if (cur && complete(cur)) slots.insert(slots.begin(), {cur_key, cur});
    else if (cur) destroy(cur);
    cur = nullptr;
    cur_key = key;
    for (size_t i = 0; i < slots.size(); ++i)
        if (slots[i].first == key) {
            cur = slots[i].second;
            slots.erase(slots.begin() + i);
            break;
        }
    while ((int)slots.size() > std::max(cap - 1, 0)) {  // least recently used goes first
        destroy(slots.back().second);
        slots.pop_back();
    }
    return usable();
}

extern "C" int edmp_unet_slot(edmp_ctx* ctx, uint64_t key) {
    if (!ctx) return EDMP_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return EDMP_ERR_HIP;
    ctx->epoch++;
    (void)prof_fold(ctx);  // pending per-op event brackets belong to the model that recorded them
    return select_slot(ctx->unet_slots, ctx->unet, ctx->unet_key, key, ctx->unet_cap, unet_destroy, unet_complete);
}

extern "C" int edmp_guide_slot(edmp_ctx* ctx, uint64_t key) {
    if (!ctx) return EDMP_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return EDMP_ERR_HIP;
    ctx->epoch++;
    return select_slot(ctx->guide_slots, ctx->guide, ctx->guide_key, key, ctx->guide_cap, [ctx](Guide* g) { guide_destroy(ctx, g); }, guide_complete);
}

extern "C" int edmp_ctx_set_stream(edmp_ctx* ctx, void* hip_stream) {
    EDMP_REQUIRE(ctx, "null ctx");
    EDMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    ctx->epoch++;
    return EDMP_OK;
}

extern "C" int edmp_ctx_synchronize(edmp_ctx* ctx) {
    EDMP_REQUIRE(ctx, "null ctx");
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    EDMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return EDMP_OK;
}

extern "C" int edmp_sampler_init(edmp_ctx* ctx, int T, double variance_thresh) {
    EDMP_REQUIRE(ctx && T >= 1, "edmp_sampler_init: bad arguments");
    ctx->epoch++;
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->sampler) {
        ctx->sampler = new Sampler();
        EDMP_HIP_CHECK(hipMalloc((void**)&ctx->sampler->sg, 14 * sizeof(double)));
    }
    Sampler* s = ctx->sampler;
    s->T = T;
    s->beta.resize(T);
    s->alpha.resize(T);
    s->alpha_bar.resize(T);
    s->c1.resize(T);
    s->sqrt_alpha.resize(T);
    // np.linspace(0, thresh, T+1)[1:]: start + i*step with step = (stop-start)/T, last element forced to stop
    const double step = variance_thresh / (double)T;
    double prod = 1.0;
    for (int i = 1; i <= T; ++i) {
        double b = (i == T) ? variance_thresh : (double)i * step;
        s->beta[i - 1] = b;
        s->alpha[i - 1] = 1.0 - b;
        prod *= s->alpha[i - 1];  // np.prod(alpha[:t]) multiplies left to right
        s->alpha_bar[i - 1] = prod;
        s->c1[i - 1] = (1.0 - s->alpha[i - 1]) / sqrt(1.0 - s->alpha_bar[i - 1]);
        s->sqrt_alpha[i - 1] = sqrt(s->alpha[i - 1]);
    }
    return EDMP_OK;
}

extern "C" int edmp_sampler_set_condition(edmp_ctx* ctx, int on) {
    EDMP_REQUIRE(ctx && ctx->sampler, "sampler not initialised");
    ctx->sampler->condition = on ? 1 : 0;
    return EDMP_OK;
}

extern "C" int edmp_sampler_read_schedule(edmp_ctx* ctx, double* beta, double* alpha, double* alpha_bar) {
    EDMP_REQUIRE(ctx && ctx->sampler, "sampler not initialised");
    Sampler* s = ctx->sampler;
    if (beta) memcpy(beta, s->beta.data(), s->T * sizeof(double));
    if (alpha) memcpy(alpha, s->alpha.data(), s->T * sizeof(double));
    if (alpha_bar) memcpy(alpha_bar, s->alpha_bar.data(), s->T * sizeof(double));
    return EDMP_OK;
}

extern "C" int edmp_psample_dev(edmp_ctx* ctx, double* X_dev, const float* eps_dev, const double* z_dev, int B, int C, int N, int t,
                                int zero_row0) {
    EDMP_REQUIRE(ctx && ctx->sampler && X_dev && eps_dev && z_dev, "edmp_psample_dev: bad arguments / sampler not initialised");
    Sampler* s = ctx->sampler;
    EDMP_REQUIRE(t >= 1 && t <= s->T, "t=%d outside 1..%d", t, s->T);
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    const int n = B * C * N;
    hipLaunchKernelGGL(psample_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, X_dev, eps_dev, z_dev, n, C * N, s->c1[t - 1],
                       s->sqrt_alpha[t - 1], s->beta[t - 1], (zero_row0 && t == 1) ? 1 : 0);
    EDMP_HIP_CHECK(hipGetLastError());
    return EDMP_OK;
}

static int check_loop_state(edmp_ctx* ctx, int B, bool guided) {
    EDMP_REQUIRE(ctx && ctx->sampler && ctx->unet, "sampler / model not initialised");
    EDMP_REQUIRE(ctx->unet->desc.input_dim == 7 || !guided, "the guide needs 7 joint channels");
    EDMP_REQUIRE(ctx->sampler->T <= ctx->unet->desc.T, "sampler T exceeds the model's time-bias table");
    EDMP_REQUIRE(B >= 1 && B <= ctx->unet->max_batch, "batch %d outside 1..%d", B, ctx->unet->max_batch);
    if (guided) {
        EDMP_REQUIRE(ctx->guide && ctx->guide->aabb && ctx->guide->row_class, "scene / rows not set");
        // the wave-per-trajectory guide kernel holds one waypoint per lane (+ start and goal): horizon <= 64; the update
        // kernel reads guidance_schedule[b][t-1] and the obstacle table has one slice per step: both must cover 1..T
        EDMP_REQUIRE(ctx->unet->desc.horizon <= 64, "guided sampling needs horizon <= 64 (one waypoint per lane), model has %d", ctx->unet->desc.horizon);
        EDMP_REQUIRE(ctx->guide->rows_T >= ctx->sampler->T, "guidance_schedule covers %d steps, the sampler runs %d", ctx->guide->rows_T, ctx->sampler->T);
        EDMP_REQUIRE(ctx->guide->T >= ctx->sampler->T, "scene tables cover %d steps, the sampler runs %d", ctx->guide->T, ctx->sampler->T);
    }
    return EDMP_OK;
}

extern "C" int edmp_step_a_dev(edmp_ctx* ctx, double* X_dev, const double* z_dev, int B, int t, const double* start, const double* goal,
                               int zero_row0, float* eps_out_dev, double* xpost_out_dev) {
    int rc = check_loop_state(ctx, B, true);
    if (rc) return rc;
    EDMP_REQUIRE(X_dev && z_dev && start && goal, "null pointer");
    EDMP_REQUIRE(t >= 1 && t <= ctx->sampler->T, "t out of range");
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    const int n = B * ctx->unet->desc.input_dim * ctx->unet->desc.horizon;
    rc = ensure_sampler_scratch(ctx, n);
    if (rc) return rc;
    rc = set_startgoal(ctx, start, goal, true);
    if (rc) return rc;
    return step_a(ctx, X_dev, z_dev, B, t, zero_row0, 1, eps_out_dev, xpost_out_dev, false);
}

extern "C" int edmp_step_b_dev(edmp_ctx* ctx, double* X_dev, int B, int t, const double* start, const double* goal, double* grad_out_dev) {
    int rc = check_loop_state(ctx, B, true);
    if (rc) return rc;
    EDMP_REQUIRE(X_dev && start && goal, "null pointer");
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    rc = set_startgoal(ctx, start, goal, false);
    if (rc) return rc;
    return step_b(ctx, X_dev, B, t, 1, grad_out_dev, false);
}

extern "C" double* edmp_sumsq_ptr_dev(edmp_ctx* ctx) { return ctx ? guide_sumsq(ctx) : nullptr; }

// The reverse loop for steps t_hi .. t_lo+1.  `init`: build X_T first (noise_dev[0] or the device RNG) and condition it;
// otherwise continue from the state left in the context by the previous segment.  noise_dev points at the first draw
// this segment consumes: [X_T draw if init][z of step t_hi][z of step t_hi-1]...  X_out_dev may be NULL (segment in the
// middle of a chunked run).  A continuation segment performs no host<->device synchronisation, so a caller can draw and
// upload the next chunk of the NumPy noise stream while this one computes.
// the stream work of one denoise_loop call: no host synchronisation, no allocation (capturable into a hipGraph)
static int enqueue_loop(edmp_ctx* ctx, const double* noise_dev, bool use_rng, uint64_t seed, int B, int guided, int t_hi, int t_lo, bool init,
                        int zero_row0, double* X_out_dev) {
    Sampler* s = ctx->sampler;
    const int C = ctx->unet->desc.input_dim, N = ctx->unet->desc.horizon;
    const size_t n = (size_t)B * C * N;
    hipStream_t st = ctx->stream;
    int rc;
    if (init) {
        // X_T with start/goal conditioning                                              diffusion.py:303-307
        if (use_rng) {
            hipLaunchKernelGGL(init_state_rng_kernel, dim3((B * N + 255) / 256), dim3(256), 0, st, seed, s->X, ctx->unet->x_in, s->sg, B, C, N,
                               s->condition);
        } else {
            EDMP_HIP_CHECK(hipMemcpyAsync(s->X, noise_dev, n * sizeof(double), hipMemcpyDeviceToDevice, st));
            if (s->condition) hipLaunchKernelGGL(condition_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, s->X, B, C, N, s->sg);
            hipLaunchKernelGGL(pack_state_kernel, dim3((B * N + 255) / 256), dim3(256), 0, st, s->X, ctx->unet->x_in, B, C, N);
            noise_dev += n;
        }
    }
    for (int t = t_hi; t > t_lo; --t) {
        const double* z = use_rng ? nullptr : noise_dev + (size_t)(t_hi - t) * n;
        rc = step_a(ctx, s->X, z, B, t, zero_row0, guided, nullptr, nullptr, true, use_rng, seed);
        if (rc) return rc;
        if (s->ar_fn && guided && guided_step(t)) {
            // sharded logical batch: this rank's sum(g^2) -> the whole batch's, enqueued by the caller's collective on
            // the context's stream (stream order is the only synchronisation: no host round trip)
            timespec h0, h1;
            clock_gettime(CLOCK_MONOTONIC, &h0);
            rc = s->ar_fn(s->ar_user, (void*)st, guide_sumsq(ctx));
            clock_gettime(CLOCK_MONOTONIC, &h1);
            const uint64_t ns = (uint64_t)((h1.tv_sec - h0.tv_sec) * 1000000000ll + (h1.tv_nsec - h0.tv_nsec));
            s->ar_calls++, s->ar_ns += ns, s->ar_max_ns = ns > s->ar_max_ns ? ns : s->ar_max_ns;
            if (rc) {
                set_error("allreduce hook failed with status %d at step t=%d", rc, t);
                return EDMP_ERR_STATE;
            }
        }
        rc = step_b(ctx, s->X, B, t, guided, nullptr, true);
        if (rc) return rc;
    }
    if (X_out_dev) EDMP_HIP_CHECK(hipMemcpyAsync(X_out_dev, s->X, n * sizeof(double), hipMemcpyDeviceToDevice, st));
    return EDMP_OK;
}

static int denoise_loop(edmp_ctx* ctx, const double* noise_dev, bool use_rng, uint64_t seed, int B, const double* start, const double* goal,
                        int guided, int t_hi, int t_lo, bool init, int zero_row0, double* X_out_dev, bool allow_graph = true) {
    int rc = check_loop_state(ctx, B, guided != 0);
    if (rc) return rc;
    EDMP_REQUIRE(noise_dev || use_rng, "null noise pointer");
    Sampler* s = ctx->sampler;
    const int T = s->T;
    EDMP_REQUIRE(t_hi >= 1 && t_hi <= T && t_lo >= 0 && t_lo < t_hi, "step range %d..%d outside 1..%d", t_hi, t_lo + 1, T);
    EDMP_REQUIRE(!init || t_hi == T, "a run starts at t = T");
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    const int C = ctx->unet->desc.input_dim, N = ctx->unet->desc.horizon;
    const size_t n = (size_t)B * C * N;
    hipStream_t st = ctx->stream;
    if (init) {
        EDMP_REQUIRE(start && goal, "null start/goal");
        rc = ensure_sampler_scratch(ctx, (int)n);
        if (rc) return rc;
        rc = set_startgoal(ctx, start, goal, guided != 0);
        if (rc) return rc;
        s->run_B = B;
    } else {
        EDMP_REQUIRE(s->X && s->run_B == B, "no run in progress for batch %d (call with init first)", B);
    }
    // a caller-supplied collective is not capturable; segments of a chunked run carry a fresh noise pointer each, so a
    // captured graph would never be replayed (capture + instantiate + destroy per chunk): they are enqueued directly
    const bool graph = allow_graph && s->graph_on == 1 && !ctx->prof.on && (!s->ar_fn || sampler_hook_is_native(s));
    Sampler::GraphKey key;
    if (graph) {
        if (guided) {
            rc = guide_prepare(ctx, B, N - 2);
            if (rc) return rc;
        }
        key.noise = noise_dev, key.hook = s->ar_fn ? s->ar_user : nullptr, key.seed = seed, key.use_rng = use_rng, key.B = B, key.guided = guided, key.t_hi = t_hi, key.t_lo = t_lo;
        key.init = init, key.zero_row0 = zero_row0, key.condition = s->condition, key.epoch = ctx->epoch;
        if (s->gexec && key == s->gkey) {
            s->graph_replays++;
            EDMP_HIP_CHECK(hipGraphLaunch(s->gexec, st));
            if (X_out_dev) EDMP_HIP_CHECK(hipMemcpyAsync(X_out_dev, s->X, n * sizeof(double), hipMemcpyDeviceToDevice, st));
            return EDMP_OK;
        }
        if (s->gexec) {
            (void)hipGraphExecDestroy(s->gexec);
            s->gexec = nullptr;
        }
        EDMP_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    }
    // the copy-out stays outside the graph: callers hand a fresh output buffer to every call
    rc = enqueue_loop(ctx, noise_dev, use_rng, seed, B, guided, t_hi, t_lo, init, zero_row0, graph ? nullptr : X_out_dev);
    if (graph) {
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(st, &g);
        if (rc) {
            if (g) (void)hipGraphDestroy(g);
            return rc;
        }
        EDMP_HIP_CHECK(e);
        e = hipGraphInstantiate(&s->gexec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        EDMP_HIP_CHECK(e);
        s->gkey = key;
        s->graph_captures++;
        EDMP_HIP_CHECK(hipGraphLaunch(s->gexec, st));
        if (X_out_dev) EDMP_HIP_CHECK(hipMemcpyAsync(X_out_dev, s->X, n * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    return rc;
}

extern "C" int edmp_denoise_guided_dev(edmp_ctx* ctx, const double* noise_dev, int B, const double* start, const double* goal, int guided,
                                       int t_stop, int zero_row0, double* X_out_dev) {
    EDMP_REQUIRE(noise_dev && X_out_dev, "edmp_denoise_guided_dev: null pointer (use edmp_denoise_guided_rng_dev for the device noise source)");
    EDMP_REQUIRE(ctx && ctx->sampler, "sampler not initialised");
    return denoise_loop(ctx, noise_dev, false, 0, B, start, goal, guided, ctx->sampler->T, t_stop, true, zero_row0, X_out_dev);
}

extern "C" int edmp_denoise_guided_rng_dev(edmp_ctx* ctx, uint64_t seed, int B, const double* start, const double* goal, int guided, int t_stop,
                                           int zero_row0, double* X_out_dev) {
    EDMP_REQUIRE(ctx && ctx->sampler && X_out_dev, "sampler not initialised / null output");
    return denoise_loop(ctx, nullptr, true, seed, B, start, goal, guided, ctx->sampler->T, t_stop, true, zero_row0, X_out_dev);
}

extern "C" int edmp_denoise_guided_segment_dev(edmp_ctx* ctx, const double* noise_dev, int B, const double* start, const double* goal, int guided,
                                               int t_hi, int t_lo, int init, int zero_row0, double* X_out_dev) {
    EDMP_REQUIRE(ctx && ctx->sampler && noise_dev, "edmp_denoise_guided_segment_dev: bad arguments");
    return denoise_loop(ctx, noise_dev, false, 0, B, start, goal, guided, t_hi, t_lo, init != 0, zero_row0, X_out_dev, false);
}

extern "C" int edmp_sampler_set_allreduce(edmp_ctx* ctx, edmp_allreduce_fn fn, void* user) {
    EDMP_REQUIRE(ctx && ctx->sampler, "sampler not initialised");
    ctx->sampler->ar_fn = fn;
    ctx->sampler->ar_user = user;
    return EDMP_OK;
}

extern "C" int edmp_sampler_set_graph(edmp_ctx* ctx, int on) {
    EDMP_REQUIRE(ctx && ctx->sampler, "sampler not initialised");
    ctx->sampler->graph_on = on ? 1 : 0;
    return EDMP_OK;
}

extern "C" int edmp_q_sample_dev(edmp_ctx* ctx, const double* x_dev, const double* eps_dev, const int32_t* t_host, int B, int C, int N,
                                 int cumulative, int condition, double* xt_dev, double* mean_dev) {
    EDMP_REQUIRE(ctx && ctx->sampler, "sampler not initialised");
    EDMP_REQUIRE(x_dev && eps_dev && t_host && xt_dev, "edmp_q_sample_dev: null pointer");
    EDMP_REQUIRE(B >= 1 && C >= 1 && N >= 1, "edmp_q_sample_dev: bad shape (%d,%d,%d)", B, C, N);
    Sampler* s = ctx->sampler;
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    std::vector<double> coef((size_t)B * 2);
    const std::vector<double>& a = cumulative ? s->alpha_bar : s->alpha;
    for (int b = 0; b < B; ++b) {
        EDMP_REQUIRE(t_host[b] >= 1 && t_host[b] <= s->T, "row %d: timestep %d outside 1..%d", b, t_host[b], s->T);
        coef[2 * b] = sqrt(a[t_host[b] - 1]);
        coef[2 * b + 1] = sqrt(1.0 - a[t_host[b] - 1]);
    }
    if (s->qcoef_cap < B) {
        if (s->qcoef) (void)hipFree(s->qcoef);
        s->qcoef = nullptr;
        s->qcoef_cap = 0;
        EDMP_HIP_CHECK(hipMalloc((void**)&s->qcoef, (size_t)B * 2 * sizeof(double)));
        s->qcoef_cap = B;
    }
    // the previous call's kernel may still be reading qcoef, and `coef` dies with this frame: order both on the stream
    EDMP_HIP_CHECK(hipMemcpyAsync(s->qcoef, coef.data(), coef.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    EDMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const int n = B * C * N;
    hipLaunchKernelGGL(q_sample_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, x_dev, eps_dev, s->qcoef, xt_dev, mean_dev, n, C * N, N,
                       condition);
    EDMP_HIP_CHECK(hipGetLastError());
    return EDMP_OK;
}

extern "C" int edmp_rng_normal_dev(edmp_ctx* ctx, uint64_t seed, int step_index, int B, int C, int N, double* out_dev) {
    EDMP_REQUIRE(ctx && out_dev && B >= 1 && C >= 1 && C <= 8 && N >= 1 && step_index >= 0, "edmp_rng_normal_dev: bad arguments");
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(rng_normal_kernel, dim3((B * N + 255) / 256), dim3(256), 0, ctx->stream, seed, step_index, out_dev, B, C, N);
    EDMP_HIP_CHECK(hipGetLastError());
    return EDMP_OK;
}

extern "C" int edmp_prof_enable(edmp_ctx* ctx, int on) {
    EDMP_REQUIRE(ctx, "null ctx");
    ctx->prof.on = on;
    return EDMP_OK;
}

extern "C" int edmp_prof_read(edmp_ctx* ctx, double* conv_ms, int64_t* conv_launches, int reset) {
    EDMP_REQUIRE(ctx, "null ctx");
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    EDMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    Prof& p = ctx->prof;
    if (int rc = prof_fold(ctx)) return rc;
    if (conv_ms) *conv_ms = p.conv_ms;
    if (conv_launches) *conv_launches = p.conv_launches;
    if (reset) {
        p.conv_ms = 0.0;
        p.conv_launches = 0;
        std::fill(p.op_ms.begin(), p.op_ms.end(), 0.0);
        std::fill(p.op_calls.begin(), p.op_calls.end(), (int64_t)0);
    }
    return EDMP_OK;
}
