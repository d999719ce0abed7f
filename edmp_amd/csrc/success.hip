// success.hip — geometric success check of every row of the finished batch, on the GPU.
//
// Stands for (reference): RobotEnvironment.benchmark_trajectory (lib/environment.py:632-680) — the plan is executed under
// position control and check_collisions (:591-608) queries contacts between the manipulator and every spawned obstacle:
// cuboids (spawn_collision_cuboids :230-247) and TRUE cylinders (spawn_collision_cylinders :249-268, radius config[7],
// height config[8], axis local z); success = no contact (:672).  The driver tallies it per scene (infer_serial.py:94-99,
// 165-168).  pybullet is a third-party dependency absent offline ("parity unpinned"), so the criterion is geometric and
// exact on the primitives the guide uses for the robot: the 9 link boxes (lib/guide.py:243-342) in float64 modified-DH
// poses against every obstacle, at every waypoint and at `substeps` joint-space interpolated configurations per segment,
// plus the joint-limit test.  The reference only PRINTS "Joint Limits Exceeded" (:659-661); its success is num_collisions == 0
// alone (:672).  Per row the kernel therefore reports three things: first (first colliding waypoint, -1 = collision-free = the
// reference's success), within (all waypoints inside the limits) and ok = within && collision-free, the STRICTER flag; callers
// tally `first < 0` where the reference's number is meant (infer_serial.py, bench.py success_proxy.collision_free_rate) and
// report `ok` beside it.  Checker: oracle/success_oracle.py (same arithmetic, NumPy).
//
// Design: one workgroup per trajectory row, thread = configuration (waypoint i, sub-step s): 197 configurations at N = 50,
// S = 4.  Each thread walks the DH chain in f64 and tests the link boxes riding each frame against the obstacles staged in
// LDS: oriented boxes by the 15-axis separating-axis test, finite cylinders by the exact clipped-polytope test.  The first
// colliding configuration of the row is an LDS integer min — deterministic.  All f64: the decision margin of touching /
// just-separated pairs (1e-9 m in the tests) is far below f32 resolution at arm's length.
#include "common.h"
#include "guide.h"

namespace edmp {

struct Robot64 {
    double dh[7][4];   // a, d, cos(alpha), sin(alpha)
    double sf[9][12];  // static frames, row-major 3x4
    double he[9][3];   // link half extents
    double qlo[7], qhi[7];
};

constexpr double kSatEps = 1e-12;  // added to |R|: near-parallel edge pairs must not produce a null axis

// box (Ra columns = axes, ca, ha) against box: 15 candidate separating axes; touching counts as overlap
__device__ __forceinline__ bool obb_overlap(const double Ra[3][3], const double ca[3], const double ha[3], const double* __restrict__ ob) {
    // ob: R (9, row-major), c (3), h (3)
    double R[3][3], A[3][3], t[3];
    const double d0 = ob[9] - ca[0], d1 = ob[10] - ca[1], d2 = ob[11] - ca[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            R[i][j] = Ra[0][i] * ob[j] + Ra[1][i] * ob[3 + j] + Ra[2][i] * ob[6 + j];
            A[i][j] = fabs(R[i][j]) + kSatEps;
        }
        t[i] = Ra[0][i] * d0 + Ra[1][i] * d1 + Ra[2][i] * d2;
    }
    const double hb0 = ob[12], hb1 = ob[13], hb2 = ob[14];
    const double hb[3] = {hb0, hb1, hb2};
    bool sep = false;
#pragma unroll
    for (int i = 0; i < 3; ++i) sep |= fabs(t[i]) > ha[i] + (hb0 * A[i][0] + hb1 * A[i][1] + hb2 * A[i][2]);
#pragma unroll
    for (int j = 0; j < 3; ++j)
        sep |= fabs(t[0] * R[0][j] + t[1] * R[1][j] + t[2] * R[2][j]) > (ha[0] * A[0][j] + ha[1] * A[1][j] + ha[2] * A[2][j]) + hb[j];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            const double ra = ha[i1] * A[i2][j] + ha[i2] * A[i1][j];
            const double rb = hb[j1] * A[i][j2] + hb[j2] * A[i][j1];
            sep |= fabs(t[i2] * R[i1][j] - t[i1] * R[i2][j]) > ra + rb;
        }
    }
    return !sep;
}

// intersect [lo, hi] with { s : |p + s d| <= h }
__device__ __forceinline__ void clip_iv(double& lo, double& hi, double p, double d, double h) {
    if (d == 0.0) {
        if (!(fabs(p) <= h)) {
            lo = 1.0;
            hi = 0.0;
        }
        return;
    }
    double s0 = (-h - p) / d, s1 = (h - p) / d;
    if (s0 > s1) {
        const double tmp = s0;
        s0 = s1;
        s1 = tmp;
    }
    lo = fmax(lo, s0);
    hi = fmin(hi, s1);
}

__device__ __forceinline__ double seg_dist2_origin(double ax, double ay, double bx, double by) {
    const double dx = bx - ax, dy = by - ay;
    const double dd = dx * dx + dy * dy;
    const double s = dd == 0.0 ? 0.0 : fmin(1.0, fmax(0.0, -(ax * dx + ay * dy) / dd));
    const double px = ax + s * dx, py = ay + s * dy;
    return px * px + py * py;
}

// box against the finite cylinder of axis = third column of the obstacle rotation, radius ob[12] * 2 (the (r, r, h) row
// carries full extents, stored halved), half height ob[14].  In the cylinder frame the box clipped to the slab |z| <= H is
// a convex polytope P; the shapes meet iff min over P of x^2 + y^2 <= r^2: 0 if the axis line pierces P, else attained on a
// projected edge of P (12 clipped box edges + the face x cap-plane segments).  oracle/success_oracle.py, same steps.
__device__ __noinline__ bool obb_cylinder_overlap(const double Rb[3][3], const double cb[3], const double hb[3], const double* __restrict__ ob) {
    double R[3][3], t[3];
    const double d0 = cb[0] - ob[9], d1 = cb[1] - ob[10], d2 = cb[2] - ob[11];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) R[i][j] = ob[i] * Rb[0][j] + ob[3 + i] * Rb[1][j] + ob[6 + i] * Rb[2][j];
        t[i] = ob[i] * d0 + ob[3 + i] * d1 + ob[6 + i] * d2;
    }
    const double H = ob[14], rad = 2.0 * ob[12], r2 = rad * rad;
    if (fabs(t[2]) > H + hb[0] * fabs(R[2][0]) + hb[1] * fabs(R[2][1]) + hb[2] * fabs(R[2][2])) return false;
    {
        double lo = -H, hi = H;
#pragma unroll
        for (int i = 0; i < 3; ++i) clip_iv(lo, hi, -(R[0][i] * t[0] + R[1][i] * t[1] + R[2][i] * t[2]), R[2][i], hb[i]);
        if (lo <= hi) return true;
    }
    bool hit = false;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = (i + 1) % 3, k = (i + 2) % 3;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double sj = (c & 2) ? hb[j] : -hb[j], sk = (c & 1) ? hb[k] : -hb[k];
            double p0[3], d[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                p0[m] = t[m] + sj * R[m][j] + sk * R[m][k] - hb[i] * R[m][i];
                d[m] = 2.0 * hb[i] * R[m][i];
            }
            double s0 = 0.0, s1 = 1.0;
            clip_iv(s0, s1, p0[2], d[2], H);
            if (s0 <= s1) hit |= seg_dist2_origin(p0[0] + s0 * d[0], p0[1] + s0 * d[1], p0[0] + s1 * d[0], p0[1] + s1 * d[1]) <= r2;
        }
    }
    if (hit) return true;
    // the face x cap-plane segments: in face coordinates (a, b) the cut is the line nj a + nk b = e, parametrised by the
    // coordinate with the smaller normal component and solved for the other (division by the dominant component only)
#pragma unroll
    for (int cap = 0; cap < 2; ++cap) {
        const double cz = cap ? -H : H;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            int j = (i + 1) % 3, k = (i + 2) % 3;
            double nj = R[2][j], nk = R[2][k];
            if (fabs(nk) > fabs(nj)) {
                const int tj = j;
                j = k;
                k = tj;
                const double tn = nj;
                nj = nk;
                nk = tn;
            }
            if (nj == 0.0) continue;  // face parallel to the cap: its edges are box edges
            const double Rj0 = R[0][j], Rj1 = R[1][j], Rk0 = R[0][k], Rk1 = R[1][k], hj = hb[j], hk = hb[k];
            const double d = -nk / nj;
#pragma unroll
            for (int sgn = 0; sgn < 2; ++sgn) {
                const double si = sgn ? hb[i] : -hb[i];
                const double f0 = t[0] + si * R[0][i], f1 = t[1] + si * R[1][i], f2 = t[2] + si * R[2][i];
                const double p = (cz - f2) / nj;
                double lo = -hk, hi = hk;
                clip_iv(lo, hi, p, d, hj);
                if (lo <= hi) {
                    const double aa = p + d * lo, ba = p + d * hi;
                    hit |= seg_dist2_origin(f0 + aa * Rj0 + lo * Rk0, f1 + aa * Rj1 + lo * Rk1, f0 + ba * Rj0 + hi * Rk0, f1 + ba * Rj1 + hi * Rk1) <= r2;
                }
            }
        }
    }
    return hit;
}

// X (B, 7, N) f64.  flags: ok[B], first[B] (first colliding waypoint, -1 none), within[B] (all waypoints inside the limits)
__global__ __launch_bounds__(256) void success_rows_kernel(const double* __restrict__ X, int B, int N, int S, const double* __restrict__ obb,
                                                           const int32_t* __restrict__ kind, int no, Robot64 rc, int32_t* __restrict__ ok,
                                                           int32_t* __restrict__ first, int32_t* __restrict__ within) {
    __shared__ double s_ob[EDMP_MAX_OBSTACLES * 16];
    __shared__ int s_kind[EDMP_MAX_OBSTACLES];
    __shared__ int s_first, s_out;
    const int r = blockIdx.x;
    const int tid = threadIdx.x;
    for (int i = tid; i < no * 16; i += 256) s_ob[i] = obb[i];
    for (int i = tid; i < no; i += 256) s_kind[i] = kind[i];
    if (tid == 0) {
        s_first = 0x7fffffff;
        s_out = 0;
    }
    __syncthreads();
    const double* xr = X + (size_t)r * 7 * N;
    // joint limits over all waypoints (incl. the pinned start / goal columns)                 lib/environment.py:659-661
    {
        int bad = 0;
        for (int e = tid; e < 7 * N; e += 256) {
            const int j = e / N;
            const double v = xr[e];
            bad |= !(v >= rc.qlo[j] - 1e-9 && v <= rc.qhi[j] + 1e-9);
        }
        if (bad) atomicOr(&s_out, 1);
    }
    const int nc = (N - 1) * S + 1;
    for (int c = tid; c < nc; c += 256) {
        const int i = c / S, s = c - i * S;
        double q[7];
        if (s == 0) {
#pragma unroll
            for (int j = 0; j < 7; ++j) q[j] = xr[j * N + i];
        } else {
            const double f = (double)s / (double)S;
#pragma unroll
            for (int j = 0; j < 7; ++j) q[j] = (1.0 - f) * xr[j * N + i] + f * xr[j * N + i + 1];
        }
        double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        double o[3] = {0, 0, 0};
        bool hit = false;
#pragma unroll 1
        for (int j = 0; j < 7 && !hit; ++j) {
            double sq, cq;
            sincos(q[j], &sq, &cq);
            const double aa = rc.dh[j][0], dd = rc.dh[j][1], ca = rc.dh[j][2], sa = rc.dh[j][3];
            const double D[3][4] = {{cq, -sq, 0.0, aa}, {sq * ca, cq * ca, -sa, -sa * dd}, {sq * sa, cq * sa, ca, ca * dd}};
            double Rn[3][3], on[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
#pragma unroll
                for (int b = 0; b < 3; ++b) Rn[a][b] = R[a][0] * D[0][b] + R[a][1] * D[1][b] + R[a][2] * D[2][b];
                on[a] = R[a][0] * D[0][3] + R[a][1] * D[1][3] + R[a][2] * D[2][3] + o[a];
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
#pragma unroll
                for (int b = 0; b < 3; ++b) R[a][b] = Rn[a][b];
                o[a] = on[a];
            }
            const int nl = (j == 6) ? 3 : 1;  // link7, hand and finger ride the last frame               lib/guide.py:93-94
            for (int ll = 0; ll < nl && !hit; ++ll) {
                const int l = (ll == 0) ? j : 6 + ll;
                double LR[3][3], Lc[3], he[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
#pragma unroll
                    for (int b = 0; b < 3; ++b) LR[a][b] = R[a][0] * rc.sf[l][b] + R[a][1] * rc.sf[l][4 + b] + R[a][2] * rc.sf[l][8 + b];
                    Lc[a] = R[a][0] * rc.sf[l][3] + R[a][1] * rc.sf[l][7] + R[a][2] * rc.sf[l][11] + o[a];
                    he[a] = rc.he[l][a];
                }
                for (int ob = 0; ob < no && !hit; ++ob) {
                    const double* od = s_ob + ob * 16;
                    hit = s_kind[ob] == 1 ? obb_cylinder_overlap(LR, Lc, he, od) : obb_overlap(LR, Lc, he, od);
                }
            }
        }
        if (hit) atomicMin(&s_first, c);
    }
    __syncthreads();
    if (tid == 0) {
        const int fc = s_first;
        const int w = s_out ? 0 : 1;
        first[r] = (fc == 0x7fffffff) ? -1 : fc / S;
        within[r] = w;
        ok[r] = (w && fc == 0x7fffffff) ? 1 : 0;
    }
}

// counts[0] = rows ok, [1] = rows within the limits, [2] = rows without a collision, [3] = B
__global__ void count_flags_kernel(const int32_t* __restrict__ ok, const int32_t* __restrict__ first, const int32_t* __restrict__ within, int B,
                                   int32_t* __restrict__ counts) {
    __shared__ int s[3];
    if (threadIdx.x < 3) s[threadIdx.x] = 0;
    __syncthreads();
    int a = 0, w = 0, f = 0;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        a += ok[i] != 0;
        w += within[i] != 0;
        f += first[i] < 0;
    }
    atomicAdd(&s[0], a);
    atomicAdd(&s[1], w);
    atomicAdd(&s[2], f);
    __syncthreads();
    if (threadIdx.x == 0) {
        counts[0] = s[0];
        counts[1] = s[1];
        counts[2] = s[2];
        counts[3] = B;
    }
}

}  // namespace edmp

using namespace edmp;

extern "C" int edmp_scene_set_shapes(edmp_ctx* ctx, const int32_t* kind, int n_obstacles) {
    EDMP_REQUIRE(ctx && ctx->guide && ctx->guide->obb && ctx->guide->kind, "edmp_scene_set_shapes: call edmp_scene_set first");
    Guide* g = ctx->guide;
    EDMP_REQUIRE(kind && n_obstacles == g->no, "edmp_scene_set_shapes: need %d kinds (one per obstacle of the scene)", g->no);
    for (int i = 0; i < n_obstacles; ++i) EDMP_REQUIRE(kind[i] == 0 || kind[i] == 1, "obstacle %d: kind must be 0 (cuboid) or 1 (cylinder)", i);
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    EDMP_HIP_CHECK(hipMemcpyAsync(g->kind, kind, n_obstacles * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    EDMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return EDMP_OK;
}

extern "C" int edmp_success_rows_dev(edmp_ctx* ctx, const double* X_dev, int B, int N, int substeps, const double* dh_f64, int32_t* ok_dev,
                                     int32_t* first_dev, int32_t* within_dev, int32_t* counts_host) {
    EDMP_REQUIRE(ctx && ctx->guide && ctx->guide->obb, "edmp_success_rows_dev: scene not set");
    Guide* g = ctx->guide;
    EDMP_REQUIRE(X_dev && B >= 1 && N >= 2 && substeps >= 1 && substeps <= 64, "edmp_success_rows_dev: need B >= 1, N >= 2, 1 <= substeps <= 64");
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    if (g->flags_B < B) {
        ctx_release(ctx, g->flags);
        g->flags = nullptr;
        g->flags_B = 0;
        if (int rc = ctx_alloc(ctx, (void**)&g->flags, ((size_t)3 * B + 4) * sizeof(int32_t))) return rc;
        g->flags_B = B;
    }
    int32_t* ok = ok_dev ? ok_dev : g->flags;
    int32_t* first = first_dev ? first_dev : g->flags + g->flags_B;
    int32_t* within = within_dev ? within_dev : g->flags + 2 * (size_t)g->flags_B;
    int32_t* counts = g->flags + 3 * (size_t)g->flags_B;
    Robot64 rc;
    for (int j = 0; j < 7; ++j)
        for (int k = 0; k < 4; ++k) rc.dh[j][k] = dh_f64 ? dh_f64[j * 4 + k] : (double)g->rc.dh[j][k];
    for (int l = 0; l < 9; ++l) {
        for (int k = 0; k < 12; ++k) rc.sf[l][k] = (double)g->rc.sf[l][k];
        for (int k = 0; k < 3; ++k) rc.he[l][k] = (double)g->rc.he[l][k];
    }
    for (int j = 0; j < 7; ++j) {
        rc.qlo[j] = g->rc.qlo[j];
        rc.qhi[j] = g->rc.qhi[j];
    }
    hipLaunchKernelGGL(success_rows_kernel, dim3(B), dim3(256), 0, ctx->stream, X_dev, B, N, substeps, g->obb, g->kind, g->no, rc, ok, first, within);
    EDMP_HIP_CHECK(hipGetLastError());
    if (counts_host) {
        hipLaunchKernelGGL(count_flags_kernel, dim3(1), dim3(256), 0, ctx->stream, ok, first, within, B, counts);
        EDMP_HIP_CHECK(hipGetLastError());
        EDMP_HIP_CHECK(hipMemcpyAsync(counts_host, counts, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        EDMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return EDMP_OK;
}
