// guide.hip — IntersectionVolumeGuide for gfx950: forward kinematics, link-box AABBs, AABB-overlap volumes
// (per-waypoint "iv" and per-segment swept "sv") and their ANALYTIC gradient w.r.t. the joints.
//
// Replaces (reference lib/guide.py): define_obstacles :118-158, get_tf_mat :45-72, forward_kinematics :74-98,
// get_link_transform :344-352, cost :354-395, swept_volume_cost :473-537, get_gradient :597-635 (autograd there),
// choose_best_trajectory :637-653.  Design:
//   * one 64-lane wave per trajectory row, lane w = padded waypoint w (0 = start, 1..L interior, L+1 = goal);
//     the swept-volume coupling between consecutive waypoints is two wave shuffles per link (neighbour AABB down,
//     routed face coefficients back up) — no LDS, no global round trip.
//   * the obstacle AABBs of the row's guide class at step t (inflation = max(size, expansion)+clearance) come from a
//     device table built once per scene and are staged in LDS per wave.
//   * autograd is replaced by the closed form  d p / d q_i = z_i x (p - o_i)  applied to the arg-min / arg-max
//     corner of every AABB face, with torch's sub-gradient conventions: first index on corner ties
//     (torch.min/max(dim)), half/half on elementwise min/max ties, clamp(min=0) passes gradient at len >= 0.
#include "common.h"
#include "guide.h"

namespace edmp {


bool guide_complete(const Guide* g) { return g && g->aabb && g->obb && g->row_class; }

// ---- the context's device block pool (common.h: edmp_ctx::pool_free) ----------------------------------------------------------
// best fit by capacity with bounded slack, so that scene after scene of the same shape reuses the same blocks
int ctx_alloc(edmp_ctx* ctx, void** p, size_t bytes) {
    const size_t need = (std::max<size_t>(bytes, 1) + 255) / 256 * 256;
    auto it = ctx->pool_free.lower_bound(need);
    if (it != ctx->pool_free.end() && it->first <= 4 * need + 65536) {
        *p = it->second;
        ctx->pool_bytes -= it->first;
        ctx->pool_free.erase(it);
        return EDMP_OK;
    }
    *p = nullptr;
    EDMP_HIP_CHECK(hipMalloc(p, need));
    ctx->pool_cap[*p] = need;
    return EDMP_OK;
}
void ctx_release(edmp_ctx* ctx, void* p) {
    if (!p) return;
    auto it = ctx->pool_cap.find(p);
    if (it == ctx->pool_cap.end()) {  // not from the pool
        (void)hipFree(p);
        return;
    }
    constexpr size_t kPoolCapBytes = size_t(512) << 20;
    if (ctx->pool_bytes + it->second > kPoolCapBytes) {
        ctx->pool_cap.erase(it);
        (void)hipFree(p);
        return;
    }
    ctx->pool_free.insert({it->second, p});
    ctx->pool_bytes += it->second;
}
void ctx_pool_destroy(edmp_ctx* ctx) {
    for (auto& e : ctx->pool_free) (void)hipFree(e.second);
    ctx->pool_free.clear();
    ctx->pool_cap.clear();
    ctx->pool_bytes = 0;
    if (ctx->d_int) (void)hipFree(ctx->d_int);
    ctx->d_int = nullptr;
}

void guide_destroy(edmp_ctx* ctx, Guide* g) {
    if (!g) return;
    for (void* p : {(void*)g->aabb, (void*)g->row_class, (void*)g->method, (void*)g->grad_norm, (void*)g->sched, (void*)g->graw,
                    (void*)g->rowsq, (void*)g->sumsq, (void*)g->startgoal, (void*)g->vol_rows, (void*)g->obb, (void*)g->kind, (void*)g->flags})
        ctx_release(ctx, p);
    delete g;
}

// ---------------------------------------------------------------------------------------------------------------
// obstacle AABB table                                                                        lib/guide.py:118-201
// ---------------------------------------------------------------------------------------------------------------
// thread per (class, t, obstacle).  sizes inflate in f64 (numpy), corners / transform / min-max in f32 (torch).
__global__ void obstacle_table_kernel(const double* __restrict__ sizes, const float* __restrict__ tf, const double* __restrict__ clearance,
                                      const double* __restrict__ expansion, float* __restrict__ out, int G, int T, int no) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G * (T + 1) * no) return;
    int o = i % no;
    int t = (i / no) % (T + 1);
    int g = i / (no * (T + 1));
    float h[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double s = sizes[o * 3 + k];
        if (t != 0) {
            s = fmax(s, expansion[(size_t)g * T + (t - 1)]);
            s = s + clearance[(size_t)g * T + (t - 1)];
        }
        h[k] = (float)s / 2.0f;
    }
    const float* m = tf + o * 12;
    float mn[3], mx[3];
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        const float sx = ((v & 3) == 1 || (v & 3) == 2) ? 1.f : -1.f;  // x signs (-,+,+,-,-,+,+,-)
        const float sy = (v & 2) ? 1.f : -1.f;                        // y signs (-,-,+,+,-,-,+,+)
        const float sz = (v & 4) ? 1.f : -1.f;                        // z signs (-,-,-,-,+,+,+,+)
        const float vx = sx * h[0], vy = sy * h[1], vz = sz * h[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float p = m[k * 4 + 0] * vx;
            p = fmaf(m[k * 4 + 1], vy, p);
            p = fmaf(m[k * 4 + 2], vz, p);
            p = p + m[k * 4 + 3];
            if (v == 0) {
                mn[k] = p;
                mx[k] = p;
            } else {
                mn[k] = fminf(mn[k], p);
                mx[k] = fmaxf(mx[k], p);
            }
        }
    }
    float* q = out + (size_t)i * 6;
    q[0] = mn[0];
    q[1] = mn[1];
    q[2] = mn[2];
    q[3] = mx[0];
    q[4] = mx[1];
    q[5] = mx[2];
}

// ---------------------------------------------------------------------------------------------------------------
// the guide kernel
// ---------------------------------------------------------------------------------------------------------------
enum GuideMode { GM_IV_VOL = 0, GM_SV_VOL = 1, GM_GRAD = 2, GM_SV_ROWSUM = 3 };

template <class TIn>
struct GuideArgs {
    const TIn* joints;  // element (r, j, wi) at joints[(r*7 + j)*ldw + off + wi], wi in 0..L-1
    int ldw, off;
    int n, L;
    int t;              // table step
    int use_row_class;  // else class 0
    int do_clip;        // clip to joint limits in f64 before the f32 cast (diffusion.py:328)
    const int32_t* row_class;
    const float* method;  // GM_GRAD: 0 iv / 1 sv per row
    const float* aabb;    // [G][T+1][no][6]
    int T, no;
    const float* startgoal;  // [14] f32
    float* out;              // volumes / raw gradient / row sums
    double* rowsq;           // GM_GRAD: per-row sum g^2
};

struct Vec3 {
    float x, y, z;
};

__device__ __forceinline__ float hw_max(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float hw_min(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float comp(const Vec3& v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }

// corner v of link box (half extents he) under transform [R|o]: world position
__device__ __forceinline__ Vec3 corner_pos(const float R[3][3], const float o[3], const float he[3], int v) {
    const float sx = ((v & 3) == 1 || (v & 3) == 2) ? 1.f : -1.f;
    const float sy = (v & 2) ? 1.f : -1.f;
    const float sz = (v & 4) ? 1.f : -1.f;
    const float vx = sx * he[0], vy = sy * he[1], vz = sz * he[2];
    Vec3 p;
    p.x = fmaf(R[0][2], vz, fmaf(R[0][1], vy, R[0][0] * vx)) + o[0];
    p.y = fmaf(R[1][2], vz, fmaf(R[1][1], vy, R[1][0] * vx)) + o[1];
    p.z = fmaf(R[2][2], vz, fmaf(R[2][1], vy, R[2][0] * vx)) + o[2];
    return p;
}

// SPLIT = 1: one wave per row, four rows per workgroup (every mode).  SPLIT = 4 (GM_GRAD, the 125 launches of a denoise_guided call;
// round 5): a workgroup = ONE row, its four waves each take a group of links - {0,1,2}, {3,4}, {5,6}, {hand, finger} (balanced on
// obstacle loop + corner search + chain rule + the DH prefix a group has to walk) - and the per-joint partial gradients are added in
// the fixed order ((w0 + w1) + w2) + w3.  One wave per row is a latency-bound layout (1024 rows = one wave per SIMD, ~15 k dependent
// VALU instructions each: 44.5 us per launch); with four waves per SIMD the same work is throughput-bound.  The link sum is
// regrouped, so results differ from SPLIT = 1 by f32 rounding (same gates).
template <int MODE, class TIn, int SPLIT = 1>
__global__ __launch_bounds__(256, (SPLIT == 4 ? 4 : 1)) void guide_kernel(GuideArgs<TIn> a, RobotConst rc) {
    static_assert(SPLIT == 1 || (SPLIT == 4 && MODE == GM_GRAD), "the link-split layout exists for the gradient only");
    __shared__ float s_obs[SPLIT == 4 ? 1 : 4][EDMP_MAX_OBSTACLES * 6];
    __shared__ float s_g[SPLIT == 4 ? 3 : 1][SPLIT == 4 ? 7 : 1][64];  // partial gradients of waves 1..3
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = (SPLIT == 4) ? blockIdx.x : blockIdx.x * 4 + wv;
    const bool row_ok = r < a.n;
    const int rr = row_ok ? r : 0;
    const int L = a.L;
    const int no = a.no;
    // obstacle AABBs of this row's class at step t -> LDS slice of this wave (SPLIT = 4: one table for the row's four waves)
    {
        const int cls = a.use_row_class ? a.row_class[rr] : 0;
        const float* src = a.aabb + ((size_t)cls * (a.T + 1) + a.t) * no * 6;
        if (SPLIT == 4) {
            for (int i = threadIdx.x; i < no * 6; i += 256) s_obs[0][i] = src[i];
        } else {
            for (int i = lane; i < no * 6; i += 64) s_obs[wv][i] = src[i];
        }
    }
    __syncthreads();
    // (scalar loads of the wave-uniform obstacle values straight from the table instead of LDS reads were tried in round 5: as many VALU
    // instructions - an SGPR operand per instruction, the rest moved into VGPRs - and the loads' latency in the loop: not kept)
    const float* obs = s_obs[SPLIT == 4 ? 0 : wv];
    // SPLIT = 4: this wave's links and the last joint frame it needs
    const int my_jmax = (SPLIT == 4) ? (wv == 0 ? 2 : wv == 1 ? 4 : 6) : 6;

    bool sv;
    if (MODE == GM_IV_VOL) sv = false;
    else if (MODE == GM_SV_VOL || MODE == GM_SV_ROWSUM) sv = true;
    else sv = a.method[rr] != 0.0f;

    // this lane's joint vector: padded waypoint w = lane (0 start, 1..L interior, >= L+1 goal)
    const int w = lane;
    float q[7];
    {
        // all seven joint values are requested before any of them is looked at (per-joint branches would serialise seven
        // round trips to a state the previous kernel has just written), lanes outside 1..L read a clamped waypoint and
        // take start / goal instead
        const int wi = min(max(w - 1, 0), L - 1);
        TIn xr[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) xr[j] = a.joints[((size_t)rr * 7 + j) * a.ldw + a.off + wi];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            float v;
            if (a.do_clip) {
                double xd = (double)xr[j];
                xd = xd < rc.qlo[j] ? rc.qlo[j] : xd;
                xd = xd > rc.qhi[j] ? rc.qhi[j] : xd;
                v = (float)xd;
            } else {
                v = (float)xr[j];
            }
            const float vs = a.startgoal[j], vg = a.startgoal[7 + j];
            q[j] = (w == 0) ? vs : ((w > L) ? vg : v);
        }
    }
    const bool interior = (w >= 1) && (w <= L);
    const bool seg_ok = (w <= L);  // segment (w, w+1)

    // cumulative transform
    float R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    float o[3] = {0, 0, 0};
    float zax[7][3], org[7][3];
    float g[7] = {0, 0, 0, 0, 0, 0, 0};
    float rowacc = 0.f;  // GM_SV_ROWSUM

#pragma unroll
    for (int j = 0; j < 7; ++j) {
        if (SPLIT == 4 && j > my_jmax) break;  // (wave-uniform)
        // T <- T * DH_j(q_j)                                                              lib/guide.py:45-72, 92
        {
            float sq, cq;
            sq = sinf(q[j]);
            cq = cosf(q[j]);
            const float aa = rc.dh[j][0], dd = rc.dh[j][1], ca = rc.dh[j][2], sa = rc.dh[j][3];
            const float D[3][4] = {{cq, -sq, 0.f, aa}, {sq * ca, cq * ca, -sa, -sa * dd}, {sq * sa, cq * sa, ca, ca * dd}};
            float Rn[3][3], on[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int c = 0; c < 3; ++c) Rn[i][c] = fmaf(R[i][2], D[2][c], fmaf(R[i][1], D[1][c], R[i][0] * D[0][c]));
                on[i] = fmaf(R[i][2], D[2][3], fmaf(R[i][1], D[1][3], R[i][0] * D[0][3])) + o[i];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int c = 0; c < 3; ++c) R[i][c] = Rn[i][c];
                o[i] = on[i];
                zax[j][i] = Rn[i][2];
                org[j][i] = on[i];
            }
        }
        // links riding this frame: link j, plus hand (7) and finger (8) on the last frame     lib/guide.py:93-94
#pragma unroll
        for (int ll = 0; ll < 3; ++ll) {
            if (ll > 0 && j != 6) continue;
            const int l = (ll == 0) ? j : 6 + ll;
            if (SPLIT == 4 && (l < 3 ? 0 : l < 5 ? 1 : l < 7 ? 2 : 3) != wv) continue;  // another wave's link (wave-uniform)
            // link transform = T * static_frame[l]                                          lib/guide.py:350
            float LR[3][3], Lo[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    LR[i][c] = fmaf(R[i][2], rc.sf[l][8 + c], fmaf(R[i][1], rc.sf[l][4 + c], R[i][0] * rc.sf[l][c]));
                Lo[i] = fmaf(R[i][2], rc.sf[l][11], fmaf(R[i][1], rc.sf[l][7], R[i][0] * rc.sf[l][3])) + o[i];
            }
            float he[3] = {rc.he[l][0], rc.he[l][1], rc.he[l][2]};
            // AABB over the 8 corners with first-index arg-min / arg-max                      lib/guide.py:370-375
            float lmin[3], lmax[3];
            int imin[3], imax[3];
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                Vec3 p = corner_pos(LR, Lo, he, v);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    float pk = comp(p, k);
                    if (v == 0) {
                        lmin[k] = pk;
                        lmax[k] = pk;
                        imin[k] = 0;
                        imax[k] = 0;
                    } else {
                        if (pk < lmin[k]) {
                            lmin[k] = pk;
                            imin[k] = v;
                        }
                        if (pk > lmax[k]) {
                            lmax[k] = pk;
                            imax[k] = v;
                        }
                    }
                }
            }
            // box used against the obstacles: own AABB (iv) or the segment AABB (sv)          lib/guide.py:513-518
            float bmin[3], bmax[3], wAmin[3], wAmax[3];
            if (sv) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    float nmin = __shfl_down(lmin[k], 1, 64);
                    float nmax = __shfl_down(lmax[k], 1, 64);
                    bmin[k] = fminf(lmin[k], nmin);
                    bmax[k] = fmaxf(lmax[k], nmax);
                    wAmin[k] = lmin[k] < nmin ? 1.f : (lmin[k] == nmin ? 0.5f : 0.f);
                    wAmax[k] = lmax[k] > nmax ? 1.f : (lmax[k] == nmax ? 0.5f : 0.f);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    bmin[k] = lmin[k];
                    bmax[k] = lmax[k];
                    wAmin[k] = 1.f;
                    wAmax[k] = 1.f;
                }
            }
            // obstacle loop: volumes and d(volume)/d(face) coefficients                      lib/guide.py:387-392
            float cmin[3] = {0, 0, 0}, cmax[3] = {0, 0, 0};
            for (int ob = 0; ob < no; ++ob) {
                const float* ab = obs + ob * 6;
                float len[3], cl[3], wlo[3], whi[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float om = ab[k], oM = ab[3 + k];
                    // (the hardware instruction itself: fmaxf / fminf make the compiler canonicalise every LDS-loaded operand first -
                    // one extra v_max_f32 x, x per obstacle value, 10 % of this loop; the table holds finite numbers)
                    const float lo = hw_max(bmin[k], om);
                    const float hi = hw_min(bmax[k], oM);
                    len[k] = hi - lo;
                    cl[k] = len[k] > 0.f ? len[k] : 0.f;
                    wlo[k] = bmin[k] > om ? 1.f : (bmin[k] == om ? 0.5f : 0.f);
                    whi[k] = bmax[k] < oM ? 1.f : (bmax[k] == oM ? 0.5f : 0.f);
                }
                const float vol = cl[0] * cl[1] * cl[2];
                if (MODE == GM_IV_VOL) {
                    if (row_ok && interior) a.out[((size_t)r * L + (w - 1)) * (9 * no) + l * no + ob] = vol;
                } else if (MODE == GM_SV_VOL) {
                    if (row_ok && seg_ok) a.out[((size_t)r * (L + 1) + w) * (9 * no) + l * no + ob] = vol;
                } else if (MODE == GM_SV_ROWSUM) {
                    if (seg_ok) rowacc += vol;
                } else {
                    const float p0 = cl[1] * cl[2], p1 = cl[0] * cl[2], p2 = cl[0] * cl[1];
                    const float pk[3] = {p0, p1, p2};
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float m = (len[k] >= 0.f) ? pk[k] : 0.f;
                        cmax[k] = fmaf(m, whi[k], cmax[k]);
                        cmin[k] = fmaf(-m, wlo[k], cmin[k]);
                    }
                }
            }
            if (MODE == GM_GRAD) {
                // route the segment-face coefficients to the waypoint that owns the face
                float tmin[3], tmax[3];
                if (sv) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float cm = seg_ok ? cmin[k] : 0.f, cM = seg_ok ? cmax[k] : 0.f;
                        const float ownA_min = cm * wAmin[k], ownA_max = cM * wAmax[k];
                        const float toB_min = cm - ownA_min, toB_max = cM - ownA_max;
                        const float fromPrev_min = __shfl_up(toB_min, 1, 64);
                        const float fromPrev_max = __shfl_up(toB_max, 1, 64);
                        tmin[k] = ownA_min + (lane > 0 ? fromPrev_min : 0.f);
                        tmax[k] = ownA_max + (lane > 0 ? fromPrev_max : 0.f);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        tmin[k] = cmin[k];
                        tmax[k] = cmax[k];
                    }
                }
                // chain rule through the arg corner of each face: d p / d q_i = z_i x (p - o_i), i <= j
#pragma unroll
                for (int k = 0; k < 3; ++k) {
#pragma unroll
                    for (int side = 0; side < 2; ++side) {
                        const float cf = side ? tmax[k] : tmin[k];
                        const int iv = side ? imax[k] : imin[k];
                        const Vec3 p = corner_pos(LR, Lo, he, iv);
#pragma unroll
                        for (int i = 0; i <= j; ++i) {
                            const float rx = p.x - org[i][0], ry = p.y - org[i][1], rz = p.z - org[i][2];
                            float d;
                            if (k == 0) d = zax[i][1] * rz - zax[i][2] * ry;
                            else if (k == 1) d = zax[i][2] * rx - zax[i][0] * rz;
                            else d = zax[i][0] * ry - zax[i][1] * rx;
                            g[i] = fmaf(cf, d, g[i]);
                        }
                    }
                }
            }
        }
    }

    if (MODE == GM_GRAD) {
        if (SPLIT == 4) {
            // partial gradients of the link groups -> wave 0, added in a fixed order
            if (wv > 0) {
#pragma unroll
                for (int i = 0; i < 7; ++i) s_g[wv - 1][i][lane] = g[i];
            }
            __syncthreads();
            if (wv > 0) return;
#pragma unroll
            for (int i = 0; i < 7; ++i) g[i] = ((g[i] + s_g[0][i][lane]) + s_g[1][i][lane]) + s_g[2][i][lane];
        }
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const float gi = interior ? g[i] : 0.f;
            if (row_ok && interior) a.out[((size_t)r * 7 + i) * L + (w - 1)] = gi;
            sq = fmaf(gi, gi, sq);
        }
        double tot = wave_sum((double)sq);
        if (row_ok && lane == 0) a.rowsq[r] = tot;
    } else if (MODE == GM_SV_ROWSUM) {
        float tot = wave_sum(rowacc);
        if (row_ok && lane == 0) a.out[r] = tot;
    }
}

// deterministic sum of the per-row partials -> one f64 (the whole-batch ||g||^2, lib/guide.py:629)
// (256 threads; the ONE summation order of sum(g^2): the stand-alone kernel and the update kernel's in-block sum share it)
__device__ __forceinline__ double block_sum_rowsq(const double* __restrict__ rowsq, int B, double* sm) {
    double s = 0.0;
    for (int i = threadIdx.x; i < B; i += 256) s += rowsq[i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    return sm[0];
}
__global__ void reduce_rowsq_kernel(const double* __restrict__ rowsq, int B, double* __restrict__ out) {
    __shared__ double sm[256];
    const double tot = block_sum_rowsq(rowsq, B, sm);
    if (threadIdx.x == 0) out[0] = tot;
}

// gradient1 = (1 - gn) * g + gn * (g / ||g||)   (lib/guide.py:627-629): f32 division, f64 mix; written as f64
__global__ void mix_gradient_kernel(const float* __restrict__ graw, const double* __restrict__ sumsq, const double* __restrict__ grad_norm,
                                    double* __restrict__ out, int B, int per_row) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * per_row) return;
    const int b = i / per_row;
    const float nrm = (float)sqrt(sumsq[0]);
    const float gv = graw[i];
    const double gn = grad_norm[b];
    out[i] = (1.0 - gn) * (double)gv + gn * (double)(gv / nrm);
}

__global__ void argmin_kernel(const float* __restrict__ v, int n, int* __restrict__ out) {
    // single wave; torch.argmin semantics (lib/guide.py:650): first index on ties, and NaN compares as the smallest
    // value, so the first NaN row wins if there is one
    int lane = threadIdx.x;
    float best = INFINITY;
    int bi = 0x7fffffff;
    bool bnan = false;
    auto better = [](float x, int i, bool xn, float b, int j, bool bn) {
        if (xn != bn) return xn;             // NaN beats every number
        if (xn) return i < j;                // both NaN: first index
        return x < b || (x == b && i < j);   // numbers: smaller, then first index
    };
    for (int i = lane; i < n; i += 64) {
        const float x = v[i];
        const bool xn = x != x;
        if (better(x, i, xn, best, bi, bnan)) {
            best = x;
            bi = i;
            bnan = xn;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        const bool on = __shfl_xor((int)bnan, o, 64) != 0;
        if (better(ob, oi, on, best, bi, bnan)) {
            best = ob;
            bi = oi;
            bnan = on;
        }
    }
    if (lane == 0) out[0] = (bi == 0x7fffffff) ? 0 : bi;
}

// *realloc (optional) is set when the blocks were replaced: captured graphs baked the old pointers in (the pool can hand
// the SAME graw block back while rowsq / vol_rows move, so comparing one pointer is not enough)
static int ensure_scratch(edmp_ctx* ctx, Guide* g, int B, int L, bool* realloc = nullptr) {
    if (realloc) *realloc = false;
    if (g->scratch_B >= B && g->scratch_L >= L) return EDMP_OK;
    if (realloc) *realloc = true;
    for (void* p : {(void*)g->graw, (void*)g->rowsq, (void*)g->vol_rows}) ctx_release(ctx, p);
    g->graw = nullptr;
    g->rowsq = nullptr;
    g->vol_rows = nullptr;
    int nb = std::max(B, g->scratch_B), nl = std::max(L, g->scratch_L);
    if (int rc = ctx_alloc(ctx, (void**)&g->graw, (size_t)nb * 7 * nl * sizeof(float))) return rc;
    if (int rc = ctx_alloc(ctx, (void**)&g->rowsq, (size_t)nb * sizeof(double))) return rc;
    if (int rc = ctx_alloc(ctx, (void**)&g->vol_rows, (size_t)nb * sizeof(float))) return rc;
    g->scratch_B = nb;
    g->scratch_L = nl;
    return EDMP_OK;
}

static int upload_startgoal(edmp_ctx* ctx, const float* start, const float* goal) {
    Guide* g = ctx->guide;
    float sg[14];
    for (int i = 0; i < 7; ++i) {
        sg[i] = start ? start[i] : 0.f;
        sg[7 + i] = goal ? goal[i] : 0.f;
    }
    EDMP_HIP_CHECK(hipMemcpyAsync(g->startgoal, sg, sizeof(sg), hipMemcpyHostToDevice, ctx->stream));
    // the host array is on the stack: make the copy complete before returning (pageable-memory copies are staged
    // synchronously by the runtime, but do not rely on it)
    EDMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return EDMP_OK;
}

template <int MODE, class TIn>
static int launch_guide(edmp_ctx* ctx, const TIn* joints, int ldw, int off, int n, int L, int t, int use_row_class, int do_clip, float* out,
                        double* rowsq) {
    Guide* g = ctx->guide;
    GuideArgs<TIn> a;
    a.joints = joints;
    a.ldw = ldw;
    a.off = off;
    a.n = n;
    a.L = L;
    a.t = t;
    a.use_row_class = use_row_class;
    a.do_clip = do_clip;
    a.row_class = g->row_class;
    a.method = g->method;
    a.aabb = g->aabb;
    a.T = g->T;
    a.no = g->no;
    a.startgoal = g->startgoal;
    a.out = out;
    a.rowsq = rowsq;
    hipStream_t st = ctx->stream;
    if constexpr (MODE == GM_GRAD) {
        // the gradient: four waves per row, one per link group (round 5: 44 -> 28 us per launch against one wave per row)
        hipLaunchKernelGGL((guide_kernel<MODE, TIn, 4>), dim3(n), dim3(256), 0, st, a, g->rc);
    } else {
        hipLaunchKernelGGL((guide_kernel<MODE, TIn, 1>), dim3((n + 3) / 4), dim3(256), 0, st, a, g->rc);
    }
    EDMP_HIP_CHECK(hipGetLastError());
    return EDMP_OK;
}

// entry points used by sampler.hip
int guide_prepare(edmp_ctx* ctx, int B, int L) {  // allocate the step scratch up front (nothing may allocate inside a graph capture)
    Guide* g = ctx->guide;
    EDMP_REQUIRE(g, "scene/rows not set");
    bool moved = false;
    int rc = ensure_scratch(ctx, g, B, L, &moved);
    if (moved) ctx->epoch++;
    return rc;
}
// reduce = false: the caller's update kernel sums the per-row partials itself (device-resident loop without a hook)
int guide_raw_gradient_from_X(edmp_ctx* ctx, const double* X_dev, int B, int N, int t, bool reduce) {
    Guide* g = ctx->guide;
    EDMP_REQUIRE(g && g->aabb && g->row_class, "scene/rows not set");
    EDMP_REQUIRE(B == g->B, "batch %d != rows set (%d)", B, g->B);
    EDMP_REQUIRE(t >= 0 && t <= g->T, "t out of range");
    int rc = ensure_scratch(ctx, g, B, N - 2);
    if (rc) return rc;
    rc = launch_guide<GM_GRAD, double>(ctx, X_dev, N, 1, B, N - 2, t, 1, 1, g->graw, g->rowsq);
    if (rc) return rc;
    if (reduce) hipLaunchKernelGGL(reduce_rowsq_kernel, dim3(1), dim3(256), 0, ctx->stream, g->rowsq, B, g->sumsq);
    EDMP_HIP_CHECK(hipGetLastError());
    return EDMP_OK;
}
const double* guide_rowsq(edmp_ctx* ctx) { return ctx->guide->rowsq; }
int guide_set_startgoal(edmp_ctx* ctx, const double* start, const double* goal) {
    float s[7], gl[7];
    for (int i = 0; i < 7; ++i) {
        s[i] = (float)start[i];
        gl[i] = (float)goal[i];
    }
    return upload_startgoal(ctx, s, gl);
}
const float* guide_graw(edmp_ctx* ctx) { return ctx->guide->graw; }
const double* guide_grad_norm(edmp_ctx* ctx) { return ctx->guide->grad_norm; }
const double* guide_sched(edmp_ctx* ctx) { return ctx->guide->sched; }
double* guide_sumsq(edmp_ctx* ctx) { return ctx->guide ? ctx->guide->sumsq : nullptr; }
int guide_rows_T(edmp_ctx* ctx) { return ctx->guide->rows_T; }

}  // namespace edmp

using namespace edmp;

// what scipy's Rotation.from_quat(q).as_matrix() computes (scalar-last, normalised, f64) — call site lib/guide.py:143
static void quat_to_matrix(const double* q, double m[3][3]) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double x = q[0] / n, y = q[1] / n, z = q[2] / n, w = q[3] / n;
    double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
    double xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
    m[0][0] = x2 - y2 - z2 + w2;
    m[0][1] = 2 * (xy - zw);
    m[0][2] = 2 * (xz + yw);
    m[1][0] = 2 * (xy + zw);
    m[1][1] = -x2 + y2 - z2 + w2;
    m[1][2] = 2 * (yz - xw);
    m[2][0] = 2 * (xz - yw);
    m[2][1] = 2 * (yz + xw);
    m[2][2] = -x2 - y2 + z2 + w2;
}

extern "C" int edmp_scene_set(edmp_ctx* ctx, const double* obstacle_config, int no, const double* clearance, const double* expansion, int G,
                              int T, const float* link_half_extents, const float* dh, const float* static_frames) {
    EDMP_REQUIRE(ctx && obstacle_config && clearance && expansion && link_half_extents && dh && static_frames, "edmp_scene_set: null argument");
    EDMP_REQUIRE(no >= 1 && no <= EDMP_MAX_OBSTACLES, "n_obstacles %d outside 1..%d", no, EDMP_MAX_OBSTACLES);
    ctx->epoch++;
    EDMP_REQUIRE(G >= 1 && T >= 1, "need at least one guide class and one step");
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    // (device blocks come from / go back to the context's pool and every copy is enqueued on the context's stream: no hipFree, no
    // null-stream copy - a scene change on one context must not wait for another context's queued loop, see common.h)
    if (!ctx->guide) {
        ctx->guide = new Guide();
        if (int rc = ctx_alloc(ctx, (void**)&ctx->guide->sumsq, sizeof(double))) return rc;
        if (int rc = ctx_alloc(ctx, (void**)&ctx->guide->startgoal, 14 * sizeof(float))) return rc;
        EDMP_HIP_CHECK(hipMemsetAsync(ctx->guide->startgoal, 0, 14 * sizeof(float), ctx->stream));
    }
    Guide* g = ctx->guide;
    // nothing enqueued may still read the tables that go back to the pool (as edmp_rows_set; hipFree used to wait for every stream)
    EDMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (void** p : {(void**)&g->aabb, (void**)&g->obb, (void**)&g->kind}) {
        ctx_release(ctx, *p);
        *p = nullptr;
    }
    g->no = no;
    g->G = G;
    g->T = T;
    memcpy(g->rc.dh, dh, sizeof(g->rc.dh));
    memcpy(g->rc.sf, static_frames, sizeof(g->rc.sf));
    memcpy(g->rc.he, link_half_extents, sizeof(g->rc.he));
    const double lo_deg[7] = {-166, -101, -166, -176, -166, -1, -166};
    const double hi_deg[7] = {166, 101, 166, -4, 166, 215, 166};
    const double pi = 3.141592653589793;  // == numpy.pi
    for (int i = 0; i < 7; ++i) {
        g->rc.qlo[i] = lo_deg[i] * (pi / 180);  // diffusion.py:282-296 evaluates deg*(np.pi/180)
        g->rc.qhi[i] = hi_deg[i] * (pi / 180);
    }
    std::vector<double> sizes(no * 3), obb((size_t)no * 16, 0.0);
    std::vector<float> tf(no * 12);
    for (int o = 0; o < no; ++o) {
        const double* c = obstacle_config + o * 10;
        double m[3][3];
        quat_to_matrix(c + 3, m);
        for (int k = 0; k < 3; ++k) {
            for (int j = 0; j < 3; ++j) {
                tf[o * 12 + k * 4 + j] = (float)m[k][j];
                obb[o * 16 + k * 3 + j] = m[k][j];
            }
            tf[o * 12 + k * 4 + 3] = (float)c[k];
            sizes[o * 3 + k] = c[7 + k];
            obb[o * 16 + 9 + k] = c[k];
            obb[o * 16 + 12 + k] = c[7 + k] / 2;  // the simulator's half extents (lib/environment.py:235)
        }
    }
    // every block first, then every enqueued copy, then ONE synchronise that all exits pass through: the host vectors and the caller's
    // arrays are read by the enqueued copies (an early return used to leave a copy from a local vector in flight)
    hipStream_t st = ctx->stream;
    const size_t n_sz = sizes.size(), n_gt = (size_t)G * T;
    double* d_in = nullptr;  // one staging block for the table kernel's inputs: sizes | clearance | expansion (f64), then the transforms (f32)
    int rc = ctx_alloc(ctx, (void**)&g->obb, obb.size() * sizeof(double));
    if (!rc) rc = ctx_alloc(ctx, (void**)&g->kind, no * sizeof(int32_t));
    if (!rc) rc = ctx_alloc(ctx, (void**)&d_in, (n_sz + 2 * n_gt) * sizeof(double) + tf.size() * sizeof(float));
    if (!rc) rc = ctx_alloc(ctx, (void**)&g->aabb, (size_t)G * (T + 1) * no * 6 * sizeof(float));
    hipError_t e = hipSuccess;
    if (!rc) {
        double *d_sizes = d_in, *d_clr = d_in + n_sz, *d_exp = d_clr + n_gt;
        float* d_tf = reinterpret_cast<float*>(d_exp + n_gt);
        e = hipMemcpyAsync(g->obb, obb.data(), obb.size() * sizeof(double), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemsetAsync(g->kind, 0, no * sizeof(int32_t), st);  // every obstacle a cuboid until edmp_scene_set_shapes says otherwise
        if (e == hipSuccess) e = hipMemcpyAsync(d_sizes, sizes.data(), n_sz * sizeof(double), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(d_tf, tf.data(), tf.size() * sizeof(float), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(d_clr, clearance, n_gt * sizeof(double), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(d_exp, expansion, n_gt * sizeof(double), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            int total = G * (T + 1) * no;
            hipLaunchKernelGGL(obstacle_table_kernel, dim3((total + 255) / 256), dim3(256), 0, st, d_sizes, d_tf, d_clr, d_exp, g->aabb, G, T, no);
            e = hipGetLastError();
        }
        const hipError_t e2 = hipStreamSynchronize(st);
        if (e == hipSuccess) e = e2;
    }
    ctx_release(ctx, d_in);
    if (rc) return rc;
    EDMP_HIP_CHECK(e);
    return EDMP_OK;
}

extern "C" int edmp_rows_set(edmp_ctx* ctx, const int32_t* row_class, const float* method, const double* grad_norm, const double* sched, int B,
                             int T) {
    EDMP_REQUIRE(ctx && ctx->guide && ctx->guide->aabb, "edmp_rows_set: call edmp_scene_set first");
    EDMP_REQUIRE(row_class && method && grad_norm && sched && B >= 1, "edmp_rows_set: null argument");
    EDMP_REQUIRE(T == ctx->guide->T, "edmp_rows_set: guidance_schedule has %d steps, the scene tables %d (the reference indexes both with t-1)", T, ctx->guide->T);
    ctx->epoch++;
    Guide* g = ctx->guide;
    for (int i = 0; i < B; ++i) {
        EDMP_REQUIRE(row_class[i] >= 0 && row_class[i] < g->G, "row %d: class %d outside 0..%d", i, row_class[i], g->G - 1);
        EDMP_REQUIRE(method[i] == 0.0f || method[i] == 1.0f, "row %d: guidance_method must be 0 (iv) or 1 (sv)", i);
    }
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    EDMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // nothing enqueued still reads the arrays that are replaced
    for (void* p : {(void*)g->row_class, (void*)g->method, (void*)g->grad_norm, (void*)g->sched}) ctx_release(ctx, p);
    g->row_class = nullptr;
    g->method = nullptr;
    g->grad_norm = nullptr;
    g->sched = nullptr;
    if (int rc = ctx_alloc(ctx, (void**)&g->row_class, B * sizeof(int32_t))) return rc;
    if (int rc = ctx_alloc(ctx, (void**)&g->method, B * sizeof(float))) return rc;
    if (int rc = ctx_alloc(ctx, (void**)&g->grad_norm, B * sizeof(double))) return rc;
    if (int rc = ctx_alloc(ctx, (void**)&g->sched, (size_t)B * T * sizeof(double))) return rc;
    hipError_t e = hipMemcpyAsync(g->row_class, row_class, B * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(g->method, method, B * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(g->grad_norm, grad_norm, B * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(g->sched, sched, (size_t)B * T * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    const hipError_t e2 = hipStreamSynchronize(ctx->stream);  // the caller's arrays may go away after the call
    EDMP_HIP_CHECK(e);
    EDMP_HIP_CHECK(e2);
    g->B = B;
    g->rows_T = T;
    return EDMP_OK;
}

extern "C" int edmp_argmin_dev(edmp_ctx* ctx, const float* v_dev, int n, int* index_host) {
    EDMP_REQUIRE(ctx && v_dev && index_host && n >= 1, "edmp_argmin_dev: bad arguments");
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->d_int) EDMP_HIP_CHECK(hipMalloc((void**)&ctx->d_int, sizeof(int)));  // kept for the life of the context
    int* d_idx = ctx->d_int;
    hipLaunchKernelGGL(argmin_kernel, dim3(1), dim3(64), 0, ctx->stream, v_dev, n, d_idx);
    hipError_t e = hipMemcpyAsync(index_host, d_idx, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    EDMP_HIP_CHECK(e);
    return EDMP_OK;
}

extern "C" int edmp_scene_read_aabbs(edmp_ctx* ctx, int cls, int t, float* out_host) {
    EDMP_REQUIRE(ctx && ctx->guide && ctx->guide->aabb && out_host, "edmp_scene_read_aabbs: scene not set");
    Guide* g = ctx->guide;
    EDMP_REQUIRE(cls >= 0 && cls < g->G && t >= 0 && t <= g->T, "class/t out of range");
    EDMP_HIP_CHECK(hipMemcpyAsync(out_host, g->aabb + ((size_t)cls * (g->T + 1) + t) * g->no * 6, (size_t)g->no * 6 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    EDMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return EDMP_OK;
}

static int check_cost_args(edmp_ctx* ctx, int n, int L, int t, int use_row_class) {
    EDMP_REQUIRE(ctx && ctx->guide && ctx->guide->aabb, "scene not set");
    Guide* g = ctx->guide;
    EDMP_REQUIRE(n >= 1 && L >= 1 && L + 2 <= 64, "need 1 <= L <= 62 waypoints per row (got %d)", L);
    EDMP_REQUIRE(t >= 0 && t <= g->T, "t=%d outside 0..%d", t, g->T);
    EDMP_REQUIRE(!use_row_class || (g->row_class && n <= g->B), "use_row_class needs edmp_rows_set with >= n rows");
    return EDMP_OK;
}

extern "C" int edmp_guide_cost_dev(edmp_ctx* ctx, const float* joints_dev, int n, int L, int t, int use_row_class, float* volumes_dev) {
    int rc = check_cost_args(ctx, n, L, t, use_row_class);
    if (rc) return rc;
    EDMP_REQUIRE(joints_dev && volumes_dev, "null pointer");
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    return launch_guide<GM_IV_VOL, float>(ctx, joints_dev, L, 0, n, L, t, use_row_class, 0, volumes_dev, nullptr);
}

extern "C" int edmp_guide_swept_cost_dev(edmp_ctx* ctx, const float* joints_dev, int n, int L, int t, int use_row_class, const float* start,
                                         const float* goal, float* volumes_dev) {
    int rc = check_cost_args(ctx, n, L, t, use_row_class);
    if (rc) return rc;
    EDMP_REQUIRE(joints_dev && volumes_dev && start && goal, "null pointer");
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    rc = upload_startgoal(ctx, start, goal);
    if (rc) return rc;
    return launch_guide<GM_SV_VOL, float>(ctx, joints_dev, L, 0, n, L, t, use_row_class, 0, volumes_dev, nullptr);
}

extern "C" int edmp_guide_gradient_dev(edmp_ctx* ctx, const double* joints_dev, int B, int L, const double* start, const double* goal, int t,
                                       double* grad_dev, double* sumsq_dev) {
    int rc = check_cost_args(ctx, B, L, t, 1);
    if (rc) return rc;
    Guide* g = ctx->guide;
    EDMP_REQUIRE(joints_dev && grad_dev && start && goal, "null pointer");
    EDMP_REQUIRE(B == g->B, "batch %d != rows set (%d)", B, g->B);
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    rc = ensure_scratch(ctx, g, B, L);
    if (rc) return rc;
    rc = guide_set_startgoal(ctx, start, goal);
    if (rc) return rc;
    rc = launch_guide<GM_GRAD, double>(ctx, joints_dev, L, 0, B, L, t, 1, 0, g->graw, g->rowsq);
    if (rc) return rc;
    hipLaunchKernelGGL(reduce_rowsq_kernel, dim3(1), dim3(256), 0, ctx->stream, g->rowsq, B, g->sumsq);
    int total = B * 7 * L;
    hipLaunchKernelGGL(mix_gradient_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, g->graw, g->sumsq, g->grad_norm, grad_dev, B, 7 * L);
    EDMP_HIP_CHECK(hipGetLastError());
    if (sumsq_dev) EDMP_HIP_CHECK(hipMemcpyAsync(sumsq_dev, g->sumsq, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return EDMP_OK;
}

extern "C" int edmp_row_swept_volumes_dev(edmp_ctx* ctx, const double* X_dev, int B, int N, const double* start, const double* goal,
                                          float* volumes_dev, int* best_index) {
    int rc = check_cost_args(ctx, B, N - 2, 0, 0);
    if (rc) return rc;
    Guide* g = ctx->guide;
    EDMP_REQUIRE(X_dev && start && goal, "null pointer");
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    rc = ensure_scratch(ctx, g, B, N - 2);
    if (rc) return rc;
    rc = guide_set_startgoal(ctx, start, goal);
    if (rc) return rc;
    float* dst = volumes_dev ? volumes_dev : g->vol_rows;
    rc = launch_guide<GM_SV_ROWSUM, double>(ctx, X_dev, N, 1, B, N - 2, 0, 0, 0, dst, nullptr);
    if (rc) return rc;
    if (best_index) {
        int* d_idx = reinterpret_cast<int*>(g->rowsq);  // scratch reuse (>= 8 bytes)
        hipLaunchKernelGGL(argmin_kernel, dim3(1), dim3(64), 0, ctx->stream, dst, B, d_idx);
        EDMP_HIP_CHECK(hipGetLastError());
        EDMP_HIP_CHECK(hipMemcpyAsync(best_index, d_idx, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        EDMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return EDMP_OK;
}

