// wide.hip — fused Conv1dBlock of the WIDE levels (Cout 256/512, L in {2, 4, 7}), round-2 design ("weights straight
// to registers").  Included by unet.hip; replaces rcb_conv_kernel there (kept for A/B builds behind EDMP_OLD_WIDE).
//
// Conv1d(k=5, pad=2) + bias -> GroupNorm(8) -> Mish -> (+ time-bias | + residual) in ONE launch
// (reference: diffusion/models/blocks.py:22-28 Conv1dBlock, :162-164 the adds of ResidualConvolutionBlock).
//
// What round 1 measured on rcb_conv_kernel (profiles/r01_*): the K loop issued MFMAs only 79 % of the time and nearly
// all of the loss was the LDS *write* path — 3/4 of every K step's staging traffic was the weight slab (CG x 32 x taps),
// pushed through ds_write_b128 (13 issue cycles each, ~80 B/clk/CU) by the same four waves that issue the MFMAs.
// Here the weights never touch LDS:
//   * at load time every conv's weights are repacked into MFMA B-FRAGMENT order: for (32-channel output slab, 8-channel
//     K group, tap slot) one contiguous 1 KiB block [lane 0..63][4]: lane (kh = lane/32, n = lane%32) holds
//     W[tap][slab*32 + n][8*kg + 4*kh + 0..3] — exactly the four B operands of four consecutive
//     v_mfma_f32_32x32x2_f32.  A wave streams its blocks with one global_load_dwordx4 per lane per block, perfectly
//     coalesced, one whole K step ahead of use (>= 2 k cycles of cover, L2/MALL latency is < 1 k).
//   * a wave = (output slab s, K slice ks): it accumulates ALL L position tiles of its slab over its share of every
//     K chunk, so each B fragment is loaded by exactly one wave of the workgroup (no duplicate fetches), every A
//     fragment read from LDS feeds up to 5 (+1) MFMA groups, all waves run the same instruction stream, and the K-slice
//     partial tiles are summed when the epilogue reads them back from LDS.
//   * only the activations go through LDS: [L][32 samples][KC + 4] per chunk in a ring of THREE stages, so chunk k+1 is
//     already visible while chunk k is consumed: the first A fragment of the next chunk is read BEFORE the step's
//     barrier and no wave ever waits for LDS latency behind a barrier; one barrier per chunk.
//   * the step's global loads (activation chunk k+2 -> staging registers, weight fragments of chunk k+1) and the
//     ds_writes of the staged chunk ride one by one in the issue shadow of the MFMA groups.
// grid = (8 groups, ceil(B/32)); blockIdx.x = group, so one XCD's L2 serves one group's weight stream to its 32 CUs.
#pragma once
#include <type_traits>

namespace edmp {

// native 16-byte vector for register staging (a float4 STRUCT copied global -> array -> LDS stays a memcpy through
// scratch memory: SROA only promotes arrays whose elements are loaded/stored as values)
using f32x4 = __attribute__((ext_vector_type(4))) float;

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// which convolution the position-tile kernel computes (tap slot of the (output position l, input position lp) pair):
//   WK_K5   Conv1d k5 s1 p2 + GroupNorm + Mish + add (Conv1dBlock, blocks.py:22-28):      tap = lp - l + 2
//   WK_DOWN Conv1d k3 s2 p1 + bias (DownSampler's last layer, blocks.py:213):              tap = lp - 2l + 1
//   WK_UP   ConvTranspose1d k4 s2 p1 + bias, cropped (UpSampler, blocks.py:251; temporalunet.py:70-71): tap = l + 1 - 2lp
enum WideKind { WK_K5 = 0, WK_DOWN = 1, WK_UP = 2 };

template <int KIND, int CG, int LIN, bool RES, int NW>
struct WideCfg {
    static constexpr int L = LIN;                    // input positions (all staged per chunk)
    static constexpr int LOUT = (KIND == WK_K5) ? LIN : (KIND == WK_DOWN) ? (LIN - 1) / 2 + 1 : ((2 * LIN == 8 || 2 * LIN == 14 || 2 * LIN == 26) ? 2 * LIN - 1 : 2 * LIN);
    static constexpr bool GN = (KIND == WK_K5);      // GroupNorm + Mish + add epilogue (else: + bias)
    static constexpr int S = CG / 32;               // 32-channel output slabs per group
    static constexpr int KSPLIT = NW / S;            // waves sharing a slab, each with its own K slice
    static constexpr int KC = (8 * KSPLIT > 32) ? 8 * KSPLIT : 32;  // channels per staged chunk
    static constexpr int QW = KC / 8 / KSPLIT;       // 8-channel K groups per wave per chunk
    static constexpr int LDK = KC + 4;
    static constexpr int KT0 = (KIND == WK_K5 && LIN == 2) ? 1 : 0;  // first tap that can be valid
    static constexpr int NTAP = (KIND == WK_K5) ? ((LIN == 2) ? 3 : 5) : (KIND == WK_DOWN) ? 3 : 4;  // taps that can be valid
    static constexpr int NSLAB = NTAP + (RES ? 1 : 0);
    // weight slot of the pair (output tile l, input position lp), -1 if the tap does not exist
    static constexpr int slot(int l, int lp) {
        const int t = (KIND == WK_K5) ? lp - l + 2 - KT0 : (KIND == WK_DOWN) ? lp - 2 * l + 1 : l + 1 - 2 * lp;
        return (t >= 0 && t < NTAP) ? t : -1;
    }
    static constexpr int A_FL = L * 32 * LDK;        // floats per activation stage
    static constexpr int NTH = NW * 64;
    static constexpr int A_F4 = L * 32 * (KC / 4);   // float4 items per stage
    static constexpr int NA = (A_F4 + NTH - 1) / NTH;
    static constexpr int NBL = QW * NSLAB;           // weight-fragment loads per wave per chunk
    static constexpr int YS = LOUT * CG + 4;
    static constexpr int NP = KSPLIT;                // partial tiles per output element
    static constexpr int NF4 = LOUT * CG / 32;       // float4 per thread in the final pass (256 threads)
    // number of MFMA groups (4 MFMAs each) of input positions [0, lp) within one K group
    static constexpr int gbase(int lp) {
        int n = 0;
        for (int x = 0; x < lp; ++x) {
            for (int l = 0; l < LOUT; ++l)
                if (slot(l, x) >= 0) ++n;
            if (RES) ++n;
        }
        return n;
    }
    // index of the group (l, lp) among the groups of input position lp
    static constexpr int gofs(int l, int lp) {
        int n = 0;
        for (int x = 0; x < l; ++x)
            if (slot(x, lp) >= 0) ++n;
        return n;
    }
    static constexpr int GPQ = gbase(L);
    static constexpr int NG = QW * GPQ;
    static constexpr int NLOAD = NA + NBL;
    static constexpr int LPG = (NLOAD + (NG - NA) - 1) / (NG - NA);  // loads per group so that they finish before the commits start
    static constexpr long valid_pairs() { return (long)GPQ - (RES ? L : 0); }
    static constexpr size_t lds_bytes() {
        size_t a = 3 * (size_t)A_FL * sizeof(float);
        size_t y = (size_t)NP * 32 * (size_t)YS * sizeof(float);
        return a > y ? a : y;
    }
    static_assert(NG > NA, "more MFMA groups than staging items");
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    static_assert(!RES || KIND == WK_K5, "the folded residual 1x1 conv belongs to a Conv1dBlock");
    static_assert((LOUT * CG) % 32 == 0, "final pass: whole float4 columns per thread");
};

// weight-fragment stream of one conv in HBM: [Cout/32][Cin/8][NSLAB][64 lanes][4] floats (Packer::conv_frag)
template <int KIND, int CG, int LIN, bool RES, int NW>
__global__ __launch_bounds__(NW * 64) void wide_conv_kernel(RcbP p) {
    using Cf = WideCfg<KIND, CG, LIN, RES, NW>;
    constexpr int L = Cf::L, LOUT = Cf::LOUT;
    constexpr int S = Cf::S, KC = Cf::KC, QW = Cf::QW, LDK = Cf::LDK, KT0 = Cf::KT0, NTAP = Cf::NTAP, NSLAB = Cf::NSLAB;
    constexpr int A_FL = Cf::A_FL, NTH = Cf::NTH, A_F4 = Cf::A_F4, NA = Cf::NA, YS = Cf::YS, NP = Cf::NP;
    constexpr int NG = Cf::NG, LPG = Cf::LPG, NBL = Cf::NBL;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    EDMP_STAMP(0, 0)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = wave % S, ks = wave / S;
    const int co0 = blockIdx.x * CG;
    const int b0 = blockIdx.y * 32;
    const int ch1 = p.C1 / KC, ch2 = p.C2 / KC;
    const int nK = ch1 + ch2;
    const int NKG = (p.C1 + p.C2) >> 3;

    // ---- activation staging map (chunk invariant): item e = tid + k*NTH -> (position, sample row, channel quad)
    int a_g1[NA], a_g2[NA], a_l[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int e = min(tid + k * NTH, A_F4 - 1);
        const int lp = e / (32 * (KC / 4)), rem = e % (32 * (KC / 4));
        const int row = rem / (KC / 4), c4 = (rem % (KC / 4)) * 4;
        const int sb = min(b0 + row, p.B - 1);
        a_g1[k] = (sb * L + lp) * p.C1 + c4;
        a_g2[k] = (sb * L + lp) * p.C2 + c4;
        a_l[k] = lp * (32 * LDK) + row * LDK + c4;
    }
    // ---- weight fragment stream of this wave
    const float* wb = p.W + ((size_t)(blockIdx.x * S + s) * NKG) * (NSLAB * 256) + lane * 4;

    f32x16 acc[LOUT];
    f32x16 racc[RES ? L : 1];
#pragma unroll
    for (int t = 0; t < LOUT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    if constexpr (RES) {
#pragma unroll
        for (int t = 0; t < L; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) racc[t][i] = 0.0f;
    }
    // epilogue operands requested up front (they land long before they are used)
    const float bias_v = (ks == 0) ? p.bias[co0 + s * 32 + (lane & 31)] : 0.0f;
    float rbias_v = 0.0f;
    if constexpr (RES) rbias_v = (ks == 0) ? p.res_bias[co0 + s * 32 + (lane & 31)] : 0.0f;

    f32x4 ra[NA];
    float4 bA[QW][NSLAB], bB[QW][NSLAB];

    auto load_a = [&](int nc, f32x4(&r)[NA]) __attribute__((always_inline)) {
        const bool first = nc < ch1;
        const float* src = first ? p.src1 : p.src2;
        const int ci0 = (first ? nc : nc - ch1) * KC;
#pragma unroll
        for (int k = 0; k < NA; ++k) r[k] = *reinterpret_cast<const f32x4*>(src + (first ? a_g1[k] : a_g2[k]) + ci0);
    };
    auto commit_a = [&](float* st, const f32x4(&r)[NA]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NA; ++k)
            if ((A_F4 % NTH == 0) || tid + k * NTH < A_F4) *reinterpret_cast<f32x4*>(st + a_l[k]) = r[k];
    };
    auto load_b = [&](int nc, float4(&b)[QW][NSLAB]) __attribute__((always_inline)) {
        const float* w = wb + ((size_t)(nc * (KC / 8) + ks * QW)) * (NSLAB * 256);
#pragma unroll
        for (int q = 0; q < QW; ++q)
#pragma unroll
            for (int t = 0; t < NSLAB; ++t) b[q][t] = *reinterpret_cast<const float4*>(w + (q * NSLAB + t) * 256);
    };

    // ---- prologue: weights of chunk 0 in flight, activation chunks 0 and 1 staged
    load_b(0, bA);
    {
        f32x4 r1[NA];
        load_a(0, ra);
        load_a(min(1, nK - 1), r1);
        commit_a(lds, ra);
        commit_a(lds + A_FL, r1);
    }
    __syncthreads();
    EDMP_STAMP(0, 1)

    const int frag = (lane & 31) * LDK + 4 * (lane >> 5) + 8 * QW * ks;
    float4 a4 = *reinterpret_cast<const float4*>(lds + frag);

#define EDMP_W_MFMA4(ACC, B4)                                                     \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, (B4).x, ACC, 0, 0, 0);       \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, (B4).y, ACC, 0, 0, 0);       \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, (B4).z, ACC, 0, 0, 0);       \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, (B4).w, ACC, 0, 0, 0);

    // one K step: MFMAs of chunk i on stage `st` with fragments bc; fetch activation chunk i+2 and the weight fragments
    // of chunk i+1 (into bn); commit the fetched activations into stage `stw`; the first A fragment of chunk i+1 is read
    // from `stn` before the barrier.  All loop indices are compile-time constants (static_for): every register array
    // is indexed statically (a runtime index, even one that would fold after unrolling, parks the array in scratch).
    auto step = [&](int i, const float* st, const float* stn, float* stw, float4(&bc)[QW][NSLAB], float4(&bn)[QW][NSLAB]) __attribute__((always_inline)) {
        const int nca = min(i + 2, nK - 1), ncb = min(i + 1, nK - 1);
        const bool first = nca < ch1;
        const float* src = first ? p.src1 : p.src2;
        const int ci0 = (first ? nca : nca - ch1) * KC;
        const float* w = wb + ((size_t)(ncb * (KC / 8) + ks * QW)) * (NSLAB * 256);
        // side work of MFMA group G: loads [G*LPG, (G+1)*LPG) of the list (NA activation loads, then NBL weight loads);
        // the last NA groups carry one ds_write of the staged chunk each
        auto side = [&](auto gc) __attribute__((always_inline)) {
            constexpr int G = decltype(gc)::value;
            static_for<G * LPG, (G + 1) * LPG>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                if constexpr (j < NA) ra[j] = *reinterpret_cast<const f32x4*>(src + (first ? a_g1[j] : a_g2[j]) + ci0);
                else if constexpr (j < NA + NBL) bn[(j - NA) / NSLAB][(j - NA) % NSLAB] = *reinterpret_cast<const float4*>(w + (j - NA) * 256);
            });
            if constexpr (G >= NG - NA) {
                constexpr int k = G - (NG - NA);
                if ((A_F4 % NTH == 0) || tid + k * NTH < A_F4) *reinterpret_cast<f32x4*>(stw + a_l[k]) = ra[k];
            }
        };
        static_for<0, QW>([&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            static_for<0, L>([&](auto lpc) __attribute__((always_inline)) {
                constexpr int lp = decltype(lpc)::value;
                // next A fragment: (lp+1, q) | (0, q+1) | first fragment of the next chunk
                const float* an_p = (lp + 1 < L) ? st + frag + (lp + 1) * (32 * LDK) + 8 * q
                                    : (q + 1 < QW) ? st + frag + 8 * (q + 1)
                                                   : stn + frag;
                const float4 an = *reinterpret_cast<const float4*>(an_p);
                static_for<0, LOUT>([&](auto lc) __attribute__((always_inline)) {
                    constexpr int l = decltype(lc)::value;
                    if constexpr (Cf::slot(l, lp) >= 0) {
                        side(std::integral_constant<int, q * Cf::GPQ + Cf::gbase(lp) + Cf::gofs(l, lp)>{});
                        __builtin_amdgcn_sched_barrier(0);  // memory work strictly BETWEEN the four-MFMA chains, never inside one
                        EDMP_W_MFMA4(acc[l], bc[q][Cf::slot(l, lp)])
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
                if constexpr (RES) {
                    side(std::integral_constant<int, q * Cf::GPQ + Cf::gbase(lp + 1) - 1>{});
                    __builtin_amdgcn_sched_barrier(0);
                    EDMP_W_MFMA4(racc[lp], bc[q][NTAP])
                    __builtin_amdgcn_sched_barrier(0);
                }
                a4 = an;
            });
        });
        __syncthreads();
    };

    {
        int i = 0;
        int sc = 0;  // stage of chunk i
        for (; i + 1 < nK; i += 2) {
            const int s1 = (sc == 2) ? 0 : sc + 1, s2 = (s1 == 2) ? 0 : s1 + 1;
            step(i, lds + sc * A_FL, lds + s1 * A_FL, lds + s2 * A_FL, bA, bB);
            step(i + 1, lds + s1 * A_FL, lds + s2 * A_FL, lds + sc * A_FL, bB, bA);
            sc = s2;
        }
        if (i < nK) {
            const int s1 = (sc == 2) ? 0 : sc + 1, s2 = (s1 == 2) ? 0 : s1 + 1;
            step(i, lds + sc * A_FL, lds + s1 * A_FL, lds + s2 * A_FL, bA, bB);
        }
    }
#undef EDMP_W_MFMA4
    EDMP_STAMP(0, 2)

    // ---- epilogue: K-slice partial tiles (+bias in slice 0) -> LDS; then per thread (sample row, 8-column part) the
    //      partials are summed and, for a Conv1dBlock, the per-sample statistics over the whole group are reduced,
    //      normalise, Mish, add; float4 stores (the closing barrier of the last step freed the stages)
    float* Y = lds;  // [NP][32][YS]
    const int et = tid & 255;  // the final pass runs on the first 256 threads
    const int erow = et >> 3, epart = et & 7;
    const int eb = min(b0 + erow, p.B - 1);
    constexpr int NF4 = Cf::NF4;
    constexpr bool GN = Cf::GN;
    float4 g4[GN ? NF4 : 1], be4[GN ? NF4 : 1], ad4[GN ? NF4 : 1];
    if constexpr (GN) {
        if (NW == 4 || tid < 256) {
#pragma unroll
            for (int i = 0; i < NF4; ++i) {
                const int col = 4 * (epart + 8 * i);
                const int l = col / CG, ch = co0 + col % CG;
                g4[i] = *reinterpret_cast<const float4*>(p.gamma + ch);
                be4[i] = *reinterpret_cast<const float4*>(p.beta + ch);
                ad4[i] = make_float4(0.f, 0.f, 0.f, 0.f);  // one addend per launch: conv1 the time bias, conv2 the residual
                if (p.add_res) ad4[i] = *reinterpret_cast<const float4*>(p.add_res + ((size_t)eb * LOUT + l) * p.Cout + ch);
                else if (p.add_tb) ad4[i] = *reinterpret_cast<const float4*>(p.add_tb + ch);
            }
        }
    }
    float* Yw = Y + ks * (32 * YS) + s * 32 + (lane & 31);
    if constexpr (RES) {
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                Yw[row * YS + l * CG] = racc[l][r] + rbias_v;
            }
        __syncthreads();
        if ((NW == 4 || tid < 256) && b0 + erow < p.B) {
#pragma unroll
            for (int i = 0; i < NF4; ++i) {
                const int col = 4 * (epart + 8 * i);
                const int l = col / CG, ch = co0 + col % CG;
                float4 rv = *reinterpret_cast<const float4*>(Y + erow * YS + col);
#pragma unroll
                for (int q = 1; q < NP; ++q) {
                    const float4 pv = *reinterpret_cast<const float4*>(Y + q * (32 * YS) + erow * YS + col);
                    rv.x += pv.x, rv.y += pv.y, rv.z += pv.z, rv.w += pv.w;
                }
                *reinterpret_cast<float4*>(p.res_out + ((size_t)(b0 + erow) * L + l) * p.Cout + ch) = rv;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int l = 0; l < LOUT; ++l)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            Yw[row * YS + l * CG] = acc[l][r] + bias_v;
        }
    __syncthreads();
    EDMP_STAMP(0, 3)
    if (NW == 4 || tid < 256) {
        const int b = b0 + erow;
        float4 v[NF4];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            v[i] = *reinterpret_cast<const float4*>(Y + erow * YS + 4 * (epart + 8 * i));
#pragma unroll
            for (int q = 1; q < NP; ++q) {
                const float4 pv = *reinterpret_cast<const float4*>(Y + q * (32 * YS) + erow * YS + 4 * (epart + 8 * i));
                v[i].x += pv.x, v[i].y += pv.y, v[i].z += pv.z, v[i].w += pv.w;
            }
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        if constexpr (GN) {
            sum += __shfl_xor(sum, 1, 64);
            sum += __shfl_xor(sum, 2, 64);
            sum += __shfl_xor(sum, 4, 64);
            constexpr float inv_n = 1.0f / (float)(LOUT * CG);
            const float mean = sum * inv_n;
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < NF4; ++i) {
                const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
                sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
            sq += __shfl_xor(sq, 1, 64);
            sq += __shfl_xor(sq, 2, 64);
            sq += __shfl_xor(sq, 4, 64);
            const float rstd = 1.0f / sqrtf(sq * inv_n + 1e-5f);
            if (b < p.B) {
#pragma unroll
                for (int i = 0; i < NF4; ++i) {
                    const int col = 4 * (epart + 8 * i);
                    const int l = col / CG, ch = co0 + col % CG;
                    const float s0 = rstd * g4[i].x, s1 = rstd * g4[i].y, s2 = rstd * g4[i].z, s3 = rstd * g4[i].w;
                    float4 o;
                    o.x = mish_fast(v[i].x * s0 + (be4[i].x - s0 * mean)) + ad4[i].x;
                    o.y = mish_fast(v[i].y * s1 + (be4[i].y - s1 * mean)) + ad4[i].y;
                    o.z = mish_fast(v[i].z * s2 + (be4[i].z - s2 * mean)) + ad4[i].z;
                    o.w = mish_fast(v[i].w * s3 + (be4[i].w - s3 * mean)) + ad4[i].w;
                    *reinterpret_cast<float4*>(p.dst + ((size_t)b * LOUT + l) * p.Cout + ch) = o;
                }
            }
        } else if (b < p.B) {
#pragma unroll
            for (int i = 0; i < NF4; ++i) {
                const int col = 4 * (epart + 8 * i);
                const int l = col / CG, ch = co0 + col % CG;
                *reinterpret_cast<float4*>(p.dst + ((size_t)b * LOUT + l) * p.Cout + ch) = v[i];
            }
        }
    }
    EDMP_STAMP(0, 4)
}

// host: [tap][Cout][Cin] (the round-1 packing, taps 0..4 and optionally tap index 5 = the folded residual 1x1 conv) ->
// fragment stream [Cout/32][Cin/8][nslab][64][4]; slot t is tap kt0 + t for t < ntap and the residual slab for t == ntap
inline void pack_fragments(const float* w_tco_ci, int cout, int cin, int kt0, int ntap, bool res, float* out) {
    const int nslab = ntap + (res ? 1 : 0);
    const int nkg = cin / 8;
    for (int sl = 0; sl < cout / 32; ++sl)
        for (int kg = 0; kg < nkg; ++kg)
            for (int t = 0; t < nslab; ++t) {
                const int tap = (t < ntap) ? kt0 + t : 5;
                float* o = out + (((size_t)sl * nkg + kg) * nslab + t) * 256;
                for (int lane = 0; lane < 64; ++lane) {
                    const int n = lane & 31, kh = lane >> 5;
                    const float* src = w_tco_ci + ((size_t)tap * cout + sl * 32 + n) * cin + 8 * kg + 4 * kh;
                    for (int j = 0; j < 4; ++j) o[lane * 4 + j] = src[j];
                }
            }
}

template <int KIND, int CG, int LIN, bool RES, int NW>
static int launch_wide_t(const RcbP& p, hipStream_t s) {
    static bool attr_set = false;
    constexpr size_t bytes = WideCfg<KIND, CG, LIN, RES, NW>::lds_bytes();
    static_assert(bytes <= 160 * 1024, "wide conv kernel exceeds the 160 KiB LDS of a CU");
    if (!attr_set) {
        EDMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_conv_kernel<KIND, CG, LIN, RES, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        attr_set = true;
    }
    dim3 grid(p.Cout / CG, (p.B + 31) / 32);
    hipLaunchKernelGGL((wide_conv_kernel<KIND, CG, LIN, RES, NW>), grid, dim3(NW * 64), bytes, s, p);
    return EDMP_OK;
}

}  // namespace edmp
