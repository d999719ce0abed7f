// level.hip — a WHOLE UNet level of the 32/64-channel resolutions in one launch (round 2).  Included by unet.hip.
//
// At L = 50 / 25 / 13 with 32 or 64 channels a convolution is a few microseconds of MFMA work; round 1 ran these levels
// as 13 launches per forward (whole-residual-block kernels + resampling convs + the final conv) that were dominated
// by launch gaps, weight staging through LDS and GroupNorm epilogues through an LDS tile (27-47 % MFMA-busy).  Here one
// workgroup owns SB whole samples x ALL channels of the level, so nothing but the level's input and output touches HBM:
//
//   LV_DOWN      x -> RCB(Cin -> C) -> RCB(C -> C) [-> skip out] -> Conv1d k3 s2 p1 -> out      (DownSampler, blocks.py:202-220)
//   LV_UP        cat(x, skip) -> RCB(2C' -> C) -> RCB(C -> C) -> ConvTranspose1d k4 s2 p1 (cropped) -> out   (UpSampler, :240-260)
//   LV_UP_FINAL  ... -> ConvTranspose -> Conv1dBlock(C -> C) -> out          (+ final_conv.0, temporalunet.py:35; the 1x1 head
//                                                                             stays fused with the posterior step)
//   RCB = ResidualConvolutionBlock (blocks.py:137-166): Conv1dBlock + time bias, Conv1dBlock, + residual (1x1 conv of the
//   input when Cin != C, identity otherwise); Conv1dBlock = Conv1d k5 p2 -> GroupNorm(8) -> Mish (blocks.py:22-28).
//
// Activations live in LDS as zero-haloed tiles [sample][2 + L + 2][channels + 4]: a conv tap is a row shift, Conv1d's zero
// padding is the halo.  GEMM rows = (sample, position), 16 per v_mfma_f32_16x16x4_f32 tile; a wave owns one 16-channel
// output slab (x half of the samples when the level has only two slabs) for ALL of its rows, so
//   * weights stream straight from HBM/L2 into registers in MFMA B-fragment order (pack_fragments, sw = 16): no LDS
//     staging, each fragment loaded by exactly one wave, one K group ahead;
//   * every (sample, GroupNorm group) lies inside ONE wave's accumulators: statistics are reduced with DPP + two
//     cross-row shuffles, normalise + Mish + add happen in registers and the result goes straight into the next stage's
//     LDS tile - no accumulator spill, no statistics pass over LDS;
//   * consecutive MFMAs go to different accumulators (tap-major, tile-minor order; the 16x16x4 MFMA has 40 cycles of
//     dependent latency for 32 of issue).
// grid = ceil(B / SB) workgroups of 256 threads.
#pragma once
#include "params.h"
#include "wide.hip"  // static_for, pack_fragments

namespace edmp {

// (LevelMode, LevelP: unet.hip, next to the other launch-parameter structs)

template <int MODE, int C, int L, int SB, int CIN>
struct LevelCfg {
    static constexpr int KX = CIN < 16 ? 16 : CIN;   // K of the first conv: stored input channels, padded to a whole K group
    static constexpr int NSLABW = C / 16;          // 16-channel output slabs = waves along N
    static constexpr int NSUB = 4 / NSLABW;        // sample subsets = waves along M
    static constexpr int SBW = SB / NSUB;          // samples per wave
    static constexpr int GS = C / 8;               // channels per GroupNorm group
    static constexpr int ROWS = SBW * L;
    static constexpr int MT = (ROWS + 15) / 16;
    // a last row tile with <= 4 real rows runs on v_mfma_f32_4x4x1_16B_f32 (16 blocks of 4x4, K = 1: the blocks = 4 column groups x the four
    // k of a fragment component; 10.5 cycles instead of 32, tools/mfma4x4_probe.hip): 50 rows cost 3 x 32 + 10.5 cycles per K step, not 128
#ifdef EDMP_NO_SMALL_TILE  // (A/B builds: __graft_entry__._build_hip_library(("-DEDMP_NO_SMALL_TILE",)), profiles/r05_switch_ab.txt)
    static constexpr bool kSmallTiles = false;
#else
    static constexpr bool kSmallTiles = true;
#endif
    static constexpr bool small_tile(int rows, int mt) { return kSmallTiles && rows - 16 * (mt - 1) <= 4; }
    static constexpr int LOUT = (MODE == LV_DOWN) ? (L - 1) / 2 + 1 : ((2 * L == 8 || 2 * L == 14 || 2 * L == 26) ? 2 * L - 1 : 2 * L);
    static constexpr int NE = (LOUT + 1) / 2, NO = LOUT / 2;  // even / odd output positions of the transposed conv
    static constexpr int MTE = (SBW * NE + 15) / 16, MTO = (SBW * NO + 15) / 16;
    static constexpr int MTR = (MODE == LV_DOWN) ? (SBW * LOUT + 15) / 16 : MTE + MTO;
    static constexpr int MTF = (MODE == LV_UP_FINAL) ? (SBW * LOUT + 15) / 16 : 0;
    static constexpr int MTMAX = (MT > MTR ? MT : MTR) > MTF ? (MT > MTR ? MT : MTR) : MTF;
    static constexpr int RSX = KX + 4, RSC = C + 4;  // tile row strides (floats)
    static constexpr int TX_FL = SB * (L + 4) * RSX;
    static constexpr int TC_FL = SB * (L + 4) * RSC;
    static constexpr int TF_FL = (MODE == LV_UP_FINAL) ? SB * (LOUT + 4) * RSC : 0;
    static constexpr int TX_ALLOC = TX_FL > TF_FL ? TX_FL : TF_FL;  // the up-sampled tile reuses the input tile
    static constexpr size_t lds_bytes() { return ((size_t)TX_ALLOC + 2 * (size_t)TC_FL) * sizeof(float); }
    static_assert(C == 32 || C == 64, "levels of 32 or 64 channels");
    static_assert(SB % NSUB == 0 && KX % 16 == 0 && CIN % 4 == 0 && SBW <= 4, "whole samples per wave, whole K groups");
};

// sum over the lanes that share this lane's GroupNorm group (GS columns of the 16-column slab, all four row quads)
template <int GS>
__device__ __forceinline__ float lv_group_sum(float x) {
    x = dpp_xor_add<1>(x);
    x = dpp_xor_add<2>(x);
    if constexpr (GS == 8) x = dpp_xor_add<4>(x);
    x = swap16_add(x);
    x = swap32_add(x);
    return x;
}

// The work of one workgroup on one level.  INR > 0: the first INR of the CIN stored input channels already sit in this level's input
// tile at the start of `lds` (zero halos included) - put there by the previous level of a merged launch (level2_kernel) - and only
// channels INR.. are staged from p.src1 / p.src2 (INR == CIN: nothing is); OUT_LDS: the resampling conv writes its result into
// `out_tile` (the NEXT level's input tile, [sample][2 + LOUT + 2][out_rs], which lies inside this level's dead input tile and is
// cleared here once that tile is dead) instead of p.out.
template <int MODE, int C, int L, int SB, int CIN, int INR = 0, bool OUT_LDS = false>
__device__ __forceinline__ void level_body(const LevelP& p, float* lds, float* out_tile, const int out_rs, const int out_fl) {
    using Cf = LevelCfg<MODE, C, L, SB, CIN>;
    constexpr bool IN_LDS = INR == CIN;  // no input staging at all
    static_assert(!OUT_LDS || MODE == LV_DOWN || MODE == LV_UP, "the last level's output leaves through the step tail / p.out");
    static_assert(INR == 0 || INR == CIN || (CIN >= 128 && INR % (CIN / 4) == 0), "a partly resident input: whole input chunks");
    constexpr int KX = Cf::KX;
    constexpr int NSLABW = Cf::NSLABW, SBW = Cf::SBW, GS = Cf::GS, MT = Cf::MT, MTMAX = Cf::MTMAX, LOUT = Cf::LOUT;
    constexpr int RSX = Cf::RSX, RSC = Cf::RSC;
    constexpr int LVSLOT = (MODE == LV_DOWN) ? (C == 32 ? 1 : 2) : (MODE == LV_UP ? 3 : 4);  // phase-stamp slot (EDMP_STAMPS builds)
    (void)LVSLOT;
    using f4 = __attribute__((ext_vector_type(4))) float;
    float* TX = lds;
    float* TA = lds + Cf::TX_ALLOC;
    float* TB = TA + Cf::TC_FL;
    float* TF = lds;  // LV_UP_FINAL: the up-sampled activation, over the dead input tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = wave % NSLABW;     // output slab
    const int hs = (wave / NSLABW) * SBW;  // first sample (within the workgroup) of this wave's rows
    const int b0 = blockIdx.x * SB;
    const int col = s * 16 + (lane & 15);  // this lane's output channel
    const int kq4 = 4 * (lane >> 4);       // channel quad of the A / B fragments
    const int rq4 = 4 * (lane >> 4);       // first accumulator row of this lane within a tile

    EDMP_STAMP(LVSLOT, 0)
    EDMP_WG_STAMP(0)
    float4 bf[6];  // weight fragments of the next stage's first K group, requested one stage early
    // ---- level input staging.  With a wide input (the up levels: CIN = 128 / 256 concatenated channels, 51 KB per workgroup
    //      that every XCD has to pull from HBM at the same moment) the tile is staged in NCH channel chunks: chunk 0 up
    //      front, chunk c+1 requested before conv1's MFMAs on chunk c and committed behind them.
    constexpr int NCH = (CIN >= 128) ? 4 : 1;            // input chunks
    constexpr int QCH = (CIN / 4) / NCH;                 // float4 per input row per chunk
    constexpr int NITC = (SB * L * QCH + 255) / 256;     // items per thread per chunk
    static_assert((CIN / 4) % NCH == 0 && (NCH == 1 || (KX == CIN && (KX / 16) % NCH == 0)), "input chunks are whole K groups");
    auto in_load = [&](int c, int it) __attribute__((always_inline)) {
        const int i = min(tid + it * 256, SB * L * QCH - 1);
        const int r = i / QCH, q = c * QCH + (i - r * QCH);
        const int sb = r / L, l = r - sb * L;
        const int bb = min(b0 + sb, p.B - 1);
        const int c14 = p.C1 >> 2;
        return (q < c14) ? *reinterpret_cast<const float4*>(p.src1 + ((size_t)bb * L + l) * p.C1 + 4 * q)
                         : *reinterpret_cast<const float4*>(p.src2 + ((size_t)bb * L + l) * p.C2 + 4 * (q - c14));
    };
    auto in_commit = [&](int c, int it, const float4& v) __attribute__((always_inline)) {
        const int i = tid + it * 256;
        if (i < SB * L * QCH) {
            const int r = i / QCH, q = c * QCH + (i - r * QCH);
            const int sb = r / L, l = r - sb * L;
            *reinterpret_cast<float4*>(TX + (sb * (L + 4) + l + 2) * RSX + 4 * q) = v;
        }
    };
    // the first input chunk (all of the input when it is staged in one piece) is requested FIRST - it is what the first MFMA
    // waits for, and memory returns in request order - then RCB 1's first weight fragments; both land while the halos are zeroed
    constexpr int NIT = (SB * L * QCH + 255) / 256;
    constexpr int FIRSTCH = (INR > 0 && !IN_LDS) ? INR / (CIN / NCH) : 0;  // first input chunk that comes from HBM
    float4 vin[NIT];
    if constexpr (!IN_LDS) {
#pragma unroll
        for (int u = 0; u < NIT; ++u) vin[u] = in_load(FIRSTCH, u);
    }
    {
        const float* w = p.w11 + ((size_t)s * (KX / 16)) * (6 * 256) + lane * 4;
#pragma unroll
        for (int t = 0; t < 6; ++t) bf[t] = *reinterpret_cast<const float4*>(w + t * 256);
    }
    // ---- zero what the convolutions read but nobody writes: the 2 + 2 halo rows of every sample in the three tiles and
    //      the padded input channels (level 0: 8 stored channels in a 16-channel K group); then stage the level input
    {
        constexpr int HX = SB * 4 * (RSX / 4), HC = SB * 4 * (RSC / 4);  // float4 items of the halo rows
        for (int i = tid + (INR > 0 ? HX : 0); i < HX + 2 * HC; i += 256) {  // (INR > 0: the previous level cleared the whole input tile before writing into it)
            const bool inx = i < HX;
            const int j = inx ? i : (i - HX) % HC;
            const int rs4 = inx ? RSX / 4 : RSC / 4;
            const int hrow = j / rs4, q = j - hrow * rs4;           // halo row index 0..4*SB-1, float4 within the row
            const int sb = hrow >> 2, hr = hrow & 3;
            const int row = sb * (L + 4) + (hr < 2 ? hr : L + hr);  // rows 0, 1, L+2, L+3 of the sample
            float* T = inx ? TX : (i - HX < HC ? TA : TB);
            *reinterpret_cast<float4*>(T + row * (inx ? RSX : RSC) + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        constexpr int cq = CIN / 4;  // float4 per input row (C1 + C2 == CIN, checked by the launcher)
        if constexpr (4 * cq < KX && INR == 0)  // padded channels of the interior rows
            for (int i = tid; i < SB * L * (KX / 4 - cq); i += 256) {
                const int r = i / (KX / 4 - cq), q = cq + i % (KX / 4 - cq);
                *reinterpret_cast<float4*>(TX + ((r / L) * (L + 4) + r % L + 2) * RSX + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        if constexpr (!IN_LDS) {
#pragma unroll
            for (int u = 0; u < NIT; ++u) in_commit(FIRSTCH, u, vin[u]);
        }
    }
    __syncthreads();

    EDMP_STAMP(LVSLOT, 1)
    f4 acc[MTMAX], racc[MT];

    // One convolution stage on this wave's rows: acc[m] += A(tile rows of m, shifted by the tap's row offset) x W[slot]
    // for the (slot, row offset) pairs of `pairs(m)`.  `ab[m]`: float offset of this lane's A row of tile m (tap offset 0,
    // channel quad included); RS: row stride of the tile; nkg: 16-channel K groups; w: this wave's fragment stream.
    // PAIRS(pi, what): compile-time table, what = 0 slot, 1 row offset, 2 target (0 acc, 1 racc), 3 parity filter
    // (-1 all tiles, 0 tiles < MTSPLIT, 1 tiles >= MTSPLIT).
    // `bfirst`: the fragments of K group 0, requested by load_first() BEFORE the previous stage's epilogue so that their
    // memory latency hides under it
    auto load_first = [&](auto nslot_c, int nkg, const float* wstream, float4(&b)[6]) __attribute__((always_inline)) {
        constexpr int NSLOT = decltype(nslot_c)::value;
        const float* w = wstream + ((size_t)s * nkg) * (NSLOT * 256) + lane * 4;
#pragma unroll
        for (int t = 0; t < NSLOT; ++t) b[t] = *reinterpret_cast<const float4*>(w + t * 256);
    };
    auto conv_stage_range = [&](auto mtn_c, auto npair_c, auto nslot_c, auto mtsplit_c, auto small_c, auto pairs, const float* tile, int RS, int nkg, const float* wstream,
                                const int(&ab)[MTMAX], float4(&bfirst)[6], int kg0, int kg1) __attribute__((always_inline)) {
        constexpr int MTN = decltype(mtn_c)::value, NPAIR = decltype(npair_c)::value, NSLOT = decltype(nslot_c)::value, MTSPLIT = decltype(mtsplit_c)::value;
        constexpr bool SMALL = decltype(small_c)::value != 0;  // the last tile holds <= 4 rows: 4x4x1 blocks, partial sums per k quarter (small_tile_finish)
        const float* w = wstream + ((size_t)s * nkg) * (NSLOT * 256) + lane * 4;
        float4 bcur[NSLOT], bnxt[NSLOT];
#pragma unroll
        for (int t = 0; t < NSLOT; ++t) bcur[t] = bfirst[t];
        for (int kg = kg0; kg < kg1; ++kg) {
            const int kgn = min(kg + 1, nkg - 1);
#pragma unroll
            for (int t = 0; t < NSLOT; ++t) bnxt[t] = *reinterpret_cast<const float4*>(w + ((size_t)kgn * NSLOT + t) * 256);
            const float* tk = tile + 16 * kg;
            // A fragments of pair 0; inside the pair loop the fragments of the NEXT pair are requested before the current
            // pair's MFMAs are issued (software pipelining by hand: one wave per SIMD has nobody else to hide LDS latency)
            float4 a[MTMAX], an[MTMAX];
            {
                constexpr int ro0 = pairs(0, 1), par0 = pairs(0, 3);
                static_for<(par0 == 1 ? MTSPLIT : 0), (par0 == 0 ? MTSPLIT : MTN)>([&](auto mc) __attribute__((always_inline)) {
                    constexpr int m = decltype(mc)::value;
                    a[m] = *reinterpret_cast<const float4*>(tk + ab[m] + ro0 * RS);
                });
            }
            static_for<0, NPAIR>([&](auto pc) __attribute__((always_inline)) {
                constexpr int pi = decltype(pc)::value;
                constexpr int slot = pairs(pi, 0), tgt = pairs(pi, 2), par = pairs(pi, 3);
                constexpr int m_lo = (par == 1) ? MTSPLIT : 0, m_hi = (par == 0) ? MTSPLIT : MTN;
                if constexpr (pi + 1 < NPAIR) {
                    constexpr int ron = pairs(pi + 1, 1), parn = pairs(pi + 1, 3);
                    static_for<(parn == 1 ? MTSPLIT : 0), (parn == 0 ? MTSPLIT : MTN)>([&](auto mc) __attribute__((always_inline)) {
                        constexpr int m = decltype(mc)::value;
                        an[m] = *reinterpret_cast<const float4*>(tk + ab[m] + ron * RS);
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
#define EDMP_LV_ROUND(J)                                                                                     \
    static_for<m_lo, m_hi>([&](auto mc) __attribute__((always_inline)) {                                   \
        constexpr int m = decltype(mc)::value;                                                               \
        if constexpr (SMALL && m == MTN - 1) {                                                               \
            if constexpr (tgt == 0) acc[m] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m].J, bcur[slot].J, acc[m], 0, 0, 0); \
            else racc[m < MT ? m : 0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m].J, bcur[slot].J, racc[m < MT ? m : 0], 0, 0, 0); \
        } else if constexpr (tgt == 0) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].J, bcur[slot].J, acc[m], 0, 0, 0); \
        else racc[m < MT ? m : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].J, bcur[slot].J, racc[m < MT ? m : 0], 0, 0, 0); \
    });
                EDMP_LV_ROUND(x)
                EDMP_LV_ROUND(y)
                EDMP_LV_ROUND(z)
                EDMP_LV_ROUND(w)
#undef EDMP_LV_ROUND
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (pi + 1 < NPAIR) {
                    constexpr int parn = pairs(pi + 1, 3);
                    static_for<(parn == 1 ? MTSPLIT : 0), (parn == 0 ? MTSPLIT : MTN)>([&](auto mc) __attribute__((always_inline)) {
                        constexpr int m = decltype(mc)::value;
                        a[m] = an[m];
                    });
                }
            });
#pragma unroll
            for (int t = 0; t < NSLOT; ++t) bcur[t] = bnxt[t];
        }
#pragma unroll
        for (int t = 0; t < NSLOT; ++t) bfirst[t] = bcur[t];  // a following range of the same stage continues with these
    };
    // after the last K group of a stage whose last tile ran on 4x4x1 blocks: lane group q (16 lanes) holds the partial sums over the k = q
    // (mod 4) of rows 0..3 of that tile; their sum (fixed order (q0 + q1) + (q2 + q3)) lands in every group - group 0's registers then mean
    // what the 16x16x4 layout says (rows 0..3), the other groups' rows >= 4 are padding the epilogues never look at
    auto small_tile_finish = [&](f4& t) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = swap32_add(swap16_add(t[r]));
    };
    auto conv_stage = [&](auto mtn_c, auto npair_c, auto nslot_c, auto mtsplit_c, auto small_c, auto pairs, const float* tile, int RS, int nkg, const float* wstream,
                          const int(&ab)[MTMAX], float4(&bfirst)[6]) __attribute__((always_inline)) {
        conv_stage_range(mtn_c, npair_c, nslot_c, mtsplit_c, small_c, pairs, tile, RS, nkg, wstream, ab, bfirst, 0, nkg);
        if constexpr (decltype(small_c)::value != 0) small_tile_finish(acc[decltype(mtn_c)::value - 1]);
    };

    // A-row offsets of a stride-1 k5 stage over a tile with LL positions per sample and row stride RS
    auto rows_k5 = [&](auto mtn_c, auto small_c, int LL, int RS, int (&ab)[MTMAX]) __attribute__((always_inline)) {
        constexpr int MTN = decltype(mtn_c)::value;
        constexpr bool SMALL = decltype(small_c)::value != 0;
#pragma unroll
        for (int m = 0; m < MTN; ++m) {
            // (a 4x4x1-block tile: lane l feeds row l % 4 of the tile - the four column groups of a k quarter read the same rows)
            const int rho = min(16 * m + ((SMALL && m == MTN - 1) ? (lane & 3) : (lane & 15)), SBW * LL - 1);
            const int sm = rho / LL, pos = rho - sm * LL;
            ab[m] = ((hs + sm) * (LL + 4) + pos) * RS + kq4;
        }
    };

    // GroupNorm(8) + Mish (+ addend) on the accumulators of a k5 stage with LL positions per sample; `emit(m, r, sample,
    // pos, y)` receives every valid element (sample = index within the workgroup)
    // Element (m, r) of a lane sits in row rho = 16 m + r + 4 q, q = lane >> 4: over the four q it touches at most two
    // samples (rows of a sample are consecutive), known at compile time: b_lo = (16 m + r) / LL and b_hi = (16 m + r + 12) / LL.
    // Where they coincide the sample is static; otherwise ONE compare against the boundary row decides.
    auto gn_epilogue = [&](auto mtn_c, auto ll_c, const float* bias, const float* gamma, const float* beta, auto addend, auto emit) __attribute__((always_inline)) {
        constexpr int MTN = decltype(mtn_c)::value, LL = decltype(ll_c)::value;
        const float bv = bias[col], gv = gamma[col], bev = beta[col];
        constexpr float inv_n = 1.0f / (float)(LL * GS);
        float mean[SBW + 1], rstd[SBW + 1], part[SBW + 1];  // slot SBW collects the padding rows behind the last sample
#pragma unroll
        for (int b = 0; b <= SBW; ++b) part[b] = 0.f;
        static_for<0, MTN * 4>([&](auto ec) __attribute__((always_inline)) {
            constexpr int m = decltype(ec)::value / 4, r = decltype(ec)::value % 4;
            constexpr int blo = (16 * m + r) / LL < SBW ? (16 * m + r) / LL : SBW, bhi = (16 * m + r + 12) / LL < SBW ? (16 * m + r + 12) / LL : SBW;
            acc[m][r] += bv;
            if constexpr (blo == bhi) part[blo] += acc[m][r];
            else {
                const bool hi = 16 * m + r + rq4 >= bhi * LL;
                part[blo] += hi ? 0.f : acc[m][r];
                part[bhi] += hi ? acc[m][r] : 0.f;
            }
        });
#pragma unroll
        for (int b = 0; b < SBW; ++b) {
            mean[b] = lv_group_sum<GS>(part[b]) * inv_n;
            part[b] = 0.f;
        }
        mean[SBW] = 0.f;
        static_for<0, MTN * 4>([&](auto ec) __attribute__((always_inline)) {
            constexpr int m = decltype(ec)::value / 4, r = decltype(ec)::value % 4;
            constexpr int blo = (16 * m + r) / LL < SBW ? (16 * m + r) / LL : SBW, bhi = (16 * m + r + 12) / LL < SBW ? (16 * m + r + 12) / LL : SBW;
            if constexpr (blo == bhi) {
                const float d = acc[m][r] - mean[blo];
                part[blo] += d * d;
            } else {
                const bool hi = 16 * m + r + rq4 >= bhi * LL;
                const float d = acc[m][r] - (hi ? mean[bhi] : mean[blo]);
                part[blo] += hi ? 0.f : d * d;
                part[bhi] += hi ? d * d : 0.f;
            }
        });
#pragma unroll
        for (int b = 0; b < SBW; ++b) rstd[b] = 1.0f / sqrtf(lv_group_sum<GS>(part[b]) * inv_n + 1e-5f);
        rstd[SBW] = 0.f;
        // normalise + Mish two elements at a time (rows r, r + 1 of a tile: same channel; packed-fp32 instructions for all but
        // min / exp2 / rcp), each element with its own sample's statistics
        static_for<0, MTN * 2>([&](auto ec) __attribute__((always_inline)) {
            constexpr int m = decltype(ec)::value / 2, r0 = 2 * (decltype(ec)::value % 2);
            float sc2[2], sh2[2];
            int sm2[2], pos2[2];
            bool ok2[2];
            static_for<0, 2>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value, r = r0 + j;
                constexpr int blo = (16 * m + r) / LL < SBW ? (16 * m + r) / LL : SBW, bhi = (16 * m + r + 12) / LL < SBW ? (16 * m + r + 12) / LL : SBW;
                const int rho = 16 * m + r + rq4;
                const bool hi = (blo != bhi) && rho >= bhi * LL;
                sm2[j] = hi ? bhi : blo;
                pos2[j] = rho - sm2[j] * LL;
                const float mu = hi ? mean[bhi] : mean[blo], rs = hi ? rstd[bhi] : rstd[blo];
                sc2[j] = rs * gv;
                sh2[j] = bev - sc2[j] * mu;
                ok2[j] = (blo < SBW) && (bhi < SBW || !hi);  // not a padding row
            });
            constexpr int blo0 = (16 * m + r0) / LL < SBW ? (16 * m + r0) / LL : SBW;
            if constexpr (blo0 < SBW) {  // (a pair whose first row lies behind the last sample is padding altogether)
                const f32x2_t y2 = mish_fast2(f32x2_t{acc[m][r0], acc[m][r0 + 1]} * f32x2_t{sc2[0], sc2[1]} + f32x2_t{sh2[0], sh2[1]});
                if (ok2[0]) emit(m, r0, hs + sm2[0], pos2[0], y2.x + addend(m, r0, hs + sm2[0], pos2[0]));
                if (ok2[1]) emit(m, r0 + 1, hs + sm2[1], pos2[1], y2.y + addend(m, r0 + 1, hs + sm2[1], pos2[1]));
            }
        });
    };
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < MTMAX; ++m) acc[m] = f4{0.f, 0.f, 0.f, 0.f};
    };
    // (slot, row offset, target, parity) tables
    constexpr auto P_K5RES = [](int pi, int what) constexpr { return what == 0 ? pi : what == 1 ? (pi < 5 ? pi : 2) : what == 2 ? (pi < 5 ? 0 : 1) : -1; };
    constexpr auto P_K5 = [](int pi, int what) constexpr { return what == 0 ? pi : what == 1 ? pi : what == 2 ? 0 : -1; };
    constexpr auto P_UP = [](int pi, int what) constexpr {
        // even output rows (tiles < MTE): (slot 3, +0), (slot 1, +1); odd rows: (slot 2, +1), (slot 0, +2)
        constexpr int slot[4] = {3, 1, 2, 0}, ro[4] = {0, 1, 1, 2};
        return what == 0 ? slot[pi] : what == 1 ? ro[pi] : what == 2 ? 0 : (pi < 2 ? 0 : 1);
    };
    using I = std::integral_constant<int, 0>;
    (void)sizeof(I);
#define EDMP_IC(v) std::integral_constant<int, (v)>{}
    int ab[MTMAX];
    auto store_tile = [&](float* T, int LL) {  // emit into a zero-haloed [sample][LL + 4][C + 4] tile
        return [=](int, int, int sm, int pos, float y) __attribute__((always_inline)) { T[(sm * (LL + 4) + pos + 2) * RSC + col] = y; };
    };
    const auto no_add = [](int, int, int, int) __attribute__((always_inline)) { return 0.0f; };

    // ================= RCB 1: conv1 (+ residual 1x1 conv) on the level input, conv2, + residual =================
    zero_acc();
#pragma unroll
    for (int m = 0; m < MT; ++m) racc[m] = f4{0.f, 0.f, 0.f, 0.f};
    constexpr int SM5 = Cf::small_tile(Cf::ROWS, MT) ? 1 : 0;  // the k5 stages at this level's length
    rows_k5(EDMP_IC(MT), EDMP_IC(SM5), L, RSX, ab);
    if constexpr (NCH == 1) {
        conv_stage(EDMP_IC(MT), EDMP_IC(6), EDMP_IC(6), EDMP_IC(0), EDMP_IC(SM5), P_K5RES, TX, RSX, KX / 16, p.w11, ab, bf);
        if constexpr (SM5 != 0) small_tile_finish(racc[MT - 1]);
    } else {
        constexpr int KGC = (KX / 16) / NCH;  // K groups per input chunk
        static_for<0, NCH>([&](auto cc) __attribute__((always_inline)) {
            constexpr int c = decltype(cc)::value;
            constexpr int nxt = FIRSTCH + c + 1;  // the chunk fetched under this chunk's MFMAs (FIRSTCH itself was staged up front)
            float4 v[NITC];
            if constexpr (nxt < NCH) {
#pragma unroll
                for (int u = 0; u < NITC; ++u) v[u] = in_load(nxt, u);
            }
            conv_stage_range(EDMP_IC(MT), EDMP_IC(6), EDMP_IC(6), EDMP_IC(0), EDMP_IC(SM5), P_K5RES, TX, RSX, KX / 16, p.w11, ab, bf, c * KGC, (c + 1) * KGC);
            if constexpr (nxt < NCH) {
#pragma unroll
                for (int u = 0; u < NITC; ++u) in_commit(nxt, u, v[u]);
                __syncthreads();
            }
        });
        if constexpr (SM5 != 0) {
            small_tile_finish(acc[MT - 1]);
            small_tile_finish(racc[MT - 1]);
        }
    }
    load_first(EDMP_IC(5), C / 16, p.w12, bf);
    EDMP_STAMP(LVSLOT, 2)
    {
        const float tbv = p.tb1[col];
        gn_epilogue(EDMP_IC(MT), EDMP_IC(L), p.b11, p.g11, p.be11, [=](int, int, int, int) __attribute__((always_inline)) { return tbv; }, store_tile(TA, L));
    }
    __syncthreads();
    if constexpr (OUT_LDS) {
        // the input tile is dead (every wave is past conv1): clear the part the next level's input tile will occupy - its halo rows and
        // padded channels must read as zeros; several barriers lie between here and the first write into it
        for (int i = tid; i < out_fl / 4; i += 256) *reinterpret_cast<float4*>(out_tile + 4 * i) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if constexpr (MODE == LV_UP_FINAL) {
        // the input tile is dead (every wave is past conv1): clear the part the up-sampled tile will occupy, so that its halo
        // rows are zero; several barriers lie between here and the first write into it
        for (int i = tid; i < Cf::TF_FL / 4; i += 256) *reinterpret_cast<float4*>(TF + 4 * i) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    zero_acc();
    rows_k5(EDMP_IC(MT), EDMP_IC(SM5), L, RSC, ab);
    EDMP_STAMP(LVSLOT, 3)
    conv_stage(EDMP_IC(MT), EDMP_IC(5), EDMP_IC(5), EDMP_IC(0), EDMP_IC(SM5), P_K5, TA, RSC, C / 16, p.w12, ab, bf);
    load_first(EDMP_IC(5), C / 16, p.w21, bf);
    EDMP_STAMP(LVSLOT, 4)
    {
        const float rbv = p.rb1[col];
        gn_epilogue(EDMP_IC(MT), EDMP_IC(L), p.b12, p.g12, p.be12, [&](int m, int r, int, int) __attribute__((always_inline)) { return racc[m][r] + rbv; }, store_tile(TB, L));
    }
    __syncthreads();
    EDMP_STAMP(LVSLOT, 5)
    // ================= RCB 2 (identity residual) =================
    zero_acc();
    conv_stage(EDMP_IC(MT), EDMP_IC(5), EDMP_IC(5), EDMP_IC(0), EDMP_IC(SM5), P_K5, TB, RSC, C / 16, p.w21, ab, bf);
    load_first(EDMP_IC(5), C / 16, p.w22, bf);
    {  // TA is free: every wave passed the barrier behind RCB 1's conv2 epilogue; TB stays, it is the residual
        const float tbv = p.tb2[col];
        gn_epilogue(EDMP_IC(MT), EDMP_IC(L), p.b21, p.g21, p.be21, [=](int, int, int, int) __attribute__((always_inline)) { return tbv; }, store_tile(TA, L));
    }
    __syncthreads();
    zero_acc();
    conv_stage(EDMP_IC(MT), EDMP_IC(5), EDMP_IC(5), EDMP_IC(0), EDMP_IC(SM5), P_K5, TA, RSC, C / 16, p.w22, ab, bf);
    if constexpr (MODE == LV_DOWN) load_first(EDMP_IC(3), C / 16, p.wrs, bf);
    else load_first(EDMP_IC(4), C / 16, p.wrs, bf);
    __syncthreads();  // TA is rewritten below with the block output
    gn_epilogue(
        EDMP_IC(MT), EDMP_IC(L), p.b22, p.g22, p.be22, [&](int, int, int sm, int pos) __attribute__((always_inline)) { return TB[(sm * (L + 4) + pos + 2) * RSC + col]; },
        [&](int, int, int sm, int pos, float y) __attribute__((always_inline)) {
            TA[(sm * (L + 4) + pos + 2) * RSC + col] = y;
            if (p.skip_out && b0 + sm < p.B) p.skip_out[((size_t)(b0 + sm) * L + pos) * C + col] = y;
        });
    __syncthreads();
    EDMP_STAMP(LVSLOT, 6)
    // ================= resampling conv on the block output (TA) =================
    zero_acc();
    const float brv = p.brs[col];
    if constexpr (MODE == LV_DOWN) {
        constexpr int MTR = Cf::MTR;
#pragma unroll
        for (int m = 0; m < MTR; ++m) {
            const int rho = min(16 * m + (lane & 15), SBW * LOUT - 1);
            const int sm = rho / LOUT, lo = rho - sm * LOUT;
            ab[m] = ((hs + sm) * (L + 4) + 2 * lo + 1) * RSC + kq4;  // tap t reads input position 2 lo - 1 + t
        }
        conv_stage(EDMP_IC(MTR), EDMP_IC(3), EDMP_IC(3), EDMP_IC(0), EDMP_IC(0), P_K5, TA, RSC, C / 16, p.wrs, ab, bf);
#pragma unroll
        for (int m = 0; m < MTR; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rho = 16 * m + rq4 + r;
                const int sm = rho / LOUT, lo = rho - sm * LOUT;
                if constexpr (OUT_LDS) {
                    if (sm < SBW) out_tile[((hs + sm) * (LOUT + 4) + lo + 2) * out_rs + col] = acc[m][r] + brv;
                } else {
                    if (sm < SBW && b0 + hs + sm < p.B) p.out[((size_t)(b0 + hs + sm) * LOUT + lo) * C + col] = acc[m][r] + brv;
                }
            }
    } else {
        constexpr int MTE = Cf::MTE, MTR = Cf::MTR, NE = Cf::NE, NO = Cf::NO;
#pragma unroll
        for (int m = 0; m < MTR; ++m) {
            const int cnt = (m < MTE) ? NE : NO;
            const int idx = min(16 * (m < MTE ? m : m - MTE) + (lane & 15), SBW * cnt - 1);
            const int sm = idx / cnt, j = idx - sm * cnt;
            ab[m] = ((hs + sm) * (L + 4) + j + 1) * RSC + kq4;  // haloed row of input position j - 1
        }
        conv_stage(EDMP_IC(MTR), EDMP_IC(4), EDMP_IC(4), EDMP_IC(MTE), EDMP_IC(0), P_UP, TA, RSC, C / 16, p.wrs, ab, bf);
        if constexpr (MODE == LV_UP_FINAL) load_first(EDMP_IC(5), C / 16, p.wfin, bf);
#pragma unroll
        for (int m = 0; m < MTR; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cnt = (m < MTE) ? NE : NO;
                const int idx = 16 * (m < MTE ? m : m - MTE) + rq4 + r;
                const int sm = idx / cnt, j = idx - sm * cnt;
                const int lo = 2 * j + (m < MTE ? 0 : 1);
                if (sm < SBW) {
                    const float y = acc[m][r] + brv;
                    if constexpr (MODE == LV_UP_FINAL) TF[((hs + sm) * (LOUT + 4) + lo + 2) * RSC + col] = y;
                    else if constexpr (OUT_LDS) out_tile[((hs + sm) * (LOUT + 4) + lo + 2) * out_rs + col] = y;
                    else if (b0 + hs + sm < p.B) p.out[((size_t)(b0 + hs + sm) * LOUT + lo) * C + col] = y;
                }
            }
        if constexpr (MODE == LV_UP_FINAL) {
            // ================= final Conv1dBlock at the up-sampled length =================
            constexpr int MTF = Cf::MTF;
            __syncthreads();
            zero_acc();
            constexpr int SMF = Cf::small_tile(SBW * LOUT, MTF) ? 1 : 0;
            rows_k5(EDMP_IC(MTF), EDMP_IC(SMF), LOUT, RSC, ab);
            // fused tail (below): this thread's state / noise values are requested before the conv stage, they land under it
            double tail_x[8], tail_z[8];
            const int tail_sm = tid / LOUT, tail_pos = tid - tail_sm * LOUT;
            const bool tail_mine = (C == 32) && p.tail.on && tid < SB * LOUT && b0 + tail_sm < p.B;
            if (tail_mine) tail_fetch(p.tail.X, p.tail.z, p.tail.rng != 0, b0 + tail_sm, tail_pos, LOUT, p.tail.C, tail_x, tail_z);
            // ... and the head's weights go into LDS behind the output tile (224 wave-uniform scalar loads in a row would
            // serialise on the scalar cache): TA / TB are dead since the barrier above
            float* TW = TA + SB * LOUT * RSC;
            if constexpr (C == 32) {
                static_assert(SB * LOUT * RSC + 8 * C + 8 <= 2 * Cf::TC_FL, "head weights fit behind the output tile");
                if (p.tail.on) {
                    for (int i = tid; i < p.tail.C * C; i += 256) TW[i] = p.tail.w[i];
                    if (tid < 8) TW[8 * C + tid] = tid < p.tail.C ? p.tail.bias[tid] : 0.0f;
                }
            }
            conv_stage(EDMP_IC(MTF), EDMP_IC(5), EDMP_IC(5), EDMP_IC(0), EDMP_IC(SMF), P_K5, TF, RSC, C / 16, p.wfin, ab, bf);
            {
                // ONE instance of the epilogue arithmetic for both destinations (two instances may be contracted differently by
                // the compiler: the loop and the stepwise API would then differ in the last ulp).
                // Device-resident loop (tail.on): the UNet's output activation never leaves the CU.  It goes into the (dead) TA/TB
                // tiles as [sample][position][C + 4]; then one thread per (sample, waypoint) runs the tail of the reverse step -
                // final 1x1 conv, posterior step, conditioning, next UNet input (tail.h; the same code head_psample_kernel runs)
                float* TY = TA;
                const bool to_lds = (C == 32) && p.tail.on;
                float yv[MTF * 4];
                int yrow[MTF * 4];  // (sample within the workgroup) * LOUT + position, -1: padding row
#pragma unroll
                for (int e = 0; e < MTF * 4; ++e) yrow[e] = -1;
                gn_epilogue(EDMP_IC(MTF), EDMP_IC(LOUT), p.bfin, p.gfin, p.befin, no_add, [&](int m, int r, int sm, int pos, float y) __attribute__((always_inline)) {
                    yv[m * 4 + r] = y;  // (m, r are compile-time constants at every call site: registers, not scratch)
                    yrow[m * 4 + r] = sm * LOUT + pos;
                });
                if (to_lds) {
#pragma unroll
                    for (int e = 0; e < MTF * 4; ++e)
                        if (yrow[e] >= 0) TY[yrow[e] * RSC + col] = yv[e];
                } else {
#pragma unroll
                    for (int e = 0; e < MTF * 4; ++e)
                        if (yrow[e] >= 0 && b0 + yrow[e] / LOUT < p.B) p.out[((size_t)b0 * LOUT + yrow[e]) * C + col] = yv[e];
                }
                if constexpr (C == 32) {
                    static_assert(SB * LOUT * RSC <= 2 * Cf::TC_FL && SB * LOUT <= 256, "the output tile fits the two dead activation tiles, one thread per row");
                    if (to_lds) __syncthreads();  // (uniform)
                    if (to_lds && tail_mine) {
                        float4 hv[C / 4];
#pragma unroll
                        for (int q = 0; q < C / 4; ++q) hv[q] = *reinterpret_cast<const float4*>(TY + tid * RSC + 4 * q);
                        const TailP& tl = p.tail;
                        const int b = b0 + tail_sm, pos = tail_pos, i = b * LOUT + pos;
#define EDMP_TAIL(FIN, RN) head_psample_item<FIN, RN, C>(hv, tail_x, tail_z, i, b, pos, TW, TW + 8 * C, tl.X, nullptr, tl.xin, tl.sg, LOUT, tl.C, tl.c1, tl.sqrt_alpha, tl.beta, tl.zero_row0, tl.seed, tl.rng_step, tl.cond)
                        if (tl.finish) {
                            if (tl.rng) EDMP_TAIL(true, true);
                            else EDMP_TAIL(true, false);
                        } else {
                            if (tl.rng) EDMP_TAIL(false, true);
                            else EDMP_TAIL(false, false);
                        }
#undef EDMP_TAIL
                    }
                }
            }
        }
    }
    EDMP_STAMP(LVSLOT, 7)
    EDMP_WG_STAMP(1)
#undef EDMP_IC
}

template <int MODE, int C, int L, int SB, int CIN>
__global__ __launch_bounds__(256, (SB <= 2 ? 2 : 1)) void level_kernel(const float* a_src1, const float* a_src2, const float* a_w11, int a_C1, int a_C2, int a_B, LevelP pr) {
    // (leading scalar arguments = what the input staging needs: preloaded into SGPRs at wave launch, see wide_conv_kernel)
    LevelP p = pr;
    p.src1 = a_src1, p.src2 = a_src2, p.w11 = a_w11, p.C1 = a_C1, p.C2 = a_C2, p.B = a_B;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    level_body<MODE, C, L, SB, CIN>(p, lds, nullptr, 0, 0);
}

// TWO consecutive levels in one launch (round 5): level A's resampled output goes straight into level B's input tile in LDS - one
// kernel boundary, one HBM round trip of the activation between them and (that part of) level B's input staging less per reverse
// step.  Both levels run the code of level_body on the same SB samples; the workgroup's LDS is the larger of the two levels' needs.
// Instances: the two down levels of the 32 / 64-channel resolutions; the two last up levels (level B's input = level A's output
// followed by the skip tensor, which still comes from HBM).
template <int MA, int CA, int LA, int CINA, int MB, int CB, int LB, int CINB, int SB>
__global__ __launch_bounds__(256, (SB <= 2 ? 2 : 1)) void level2_kernel(const float* a_src1, const float* a_src2, const float* a_w11, int a_C1, int a_C2, int a_B, LevelP pa, LevelP pb) {
    using CfA = LevelCfg<MA, CA, LA, SB, CINA>;
    using CfB = LevelCfg<MB, CB, LB, SB, CINB>;
    static_assert(CfA::LOUT == LB, "level B runs at level A's output length");
    static_assert(CfB::KX == CINB && CINB >= CA && CfB::TX_FL <= CfA::TX_ALLOC, "level B's input tile fits inside level A's (dead) input tile");
    LevelP p = pa;
    p.src1 = a_src1, p.src2 = a_src2, p.w11 = a_w11, p.C1 = a_C1, p.C2 = a_C2, p.B = a_B;
    pb.B = a_B;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    level_body<MA, CA, LA, SB, CINA, 0, true>(p, lds, lds, CfB::RSX, CfB::TX_FL);
    __syncthreads();  // every wave has written its part of level B's input and is done with level A's tiles
    level_body<MB, CB, LB, SB, CINB, CA, false>(pb, lds, nullptr, 0, 0);
}

template <int MODE, int C, int L, int SB, int CIN>
int launch_level_t(const LevelP& p, hipStream_t s) {
    // set once per instance; launches come from several host threads (two contexts: scenes in flight, chains of one batch)
    static std::atomic<int> attr_set{0};
    EDMP_REQUIRE(p.C1 + p.C2 == CIN && p.C1 % 4 == 0 && p.C2 % 4 == 0, "level kernel built for %d stored input channels, got %d + %d", CIN, p.C1, p.C2);
    constexpr size_t bytes = LevelCfg<MODE, C, L, SB, CIN>::lds_bytes();
    static_assert(bytes <= 160 * 1024, "level kernel exceeds the 160 KiB LDS of a CU");
    if (!attr_set.load(std::memory_order_acquire)) {  // idempotent call: a second thread racing here at worst repeats it
        EDMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&level_kernel<MODE, C, L, SB, CIN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        attr_set.store(1, std::memory_order_release);
    }
    hipLaunchKernelGGL((level_kernel<MODE, C, L, SB, CIN>), dim3((p.B + SB - 1) / SB), dim3(256), bytes, s, p.src1, p.src2, p.w11, p.C1, p.C2, p.B, p);
    return EDMP_OK;
}

template <int MA, int CA, int LA, int CINA, int MB, int CB, int LB, int CINB, int SB>
int launch_level2_t(const LevelP& pa, const LevelP& pb, hipStream_t s) {
    static std::atomic<int> attr_set{0};
    EDMP_REQUIRE(pa.C1 + pa.C2 == CINA && pa.C1 % 4 == 0 && pa.C2 % 4 == 0 && pb.C1 == CA && pb.C1 + pb.C2 == CINB, "merged level kernel built for %d -> %d + %d stored input channels, got %d + %d -> %d + %d",
                 CINA, CA, CINB - CA, pa.C1, pa.C2, pb.C1, pb.C2);
    constexpr size_t ba = LevelCfg<MA, CA, LA, SB, CINA>::lds_bytes(), bb = LevelCfg<MB, CB, LB, SB, CINB>::lds_bytes();
    constexpr size_t bytes = ba > bb ? ba : bb;
    static_assert(bytes <= 160 * 1024, "merged level kernel exceeds the 160 KiB LDS of a CU");
    if (!attr_set.load(std::memory_order_acquire)) {
        EDMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&level2_kernel<MA, CA, LA, CINA, MB, CB, LB, CINB, SB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        attr_set.store(1, std::memory_order_release);
    }
    hipLaunchKernelGGL((level2_kernel<MA, CA, LA, CINA, MB, CB, LB, CINB, SB>), dim3((pa.B + SB - 1) / SB), dim3(256), bytes, s, pa.src1, pa.src2, pa.w11, pa.C1, pa.C2, pa.B, pa, pb);
    return EDMP_OK;
}

}  // namespace edmp
