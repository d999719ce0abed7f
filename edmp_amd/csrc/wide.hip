// wide.hip — the "position-tile" convolution kernel of the UNet levels with >= 128 channels (round-2 design, weights
// straight to registers).
//
//   WK_K5   Conv1d(k=5, pad=2) + bias -> GroupNorm(8) -> Mish -> (+ time-bias | + residual) in ONE launch, optionally with
//           the block's residual 1x1 conv folded in (reference: diffusion/models/blocks.py:22-28 Conv1dBlock, :162-164
//           the adds of ResidualConvolutionBlock, :147-152 residual_conv)
//   WK_DOWN Conv1d(k=3, stride 2, pad 1) + bias        (DownSampler's last layer, blocks.py:213)
//   WK_UP   ConvTranspose1d(k=4, stride 2, pad 1) + bias, cropped to 2L-1 where the reference crops (blocks.py:251,
//           temporalunet.py:70-71)
//
// A workgroup owns MS samples x CG output channels (whole GroupNorm groups) x ALL output positions; an accumulator tile
// is (one output position, MS samples) x (one slab of output channels), so taps are never materialised: the pair
// (output position l, input position lp) contributes A[lp] x W[tap(l, lp)] iff that tap exists — padding taps are never
// issued and there are no halo rows.  MS = 32 uses v_mfma_f32_32x32x2_f32 (levels with >= 256 channels), MS = 16 uses
// v_mfma_f32_16x16x4_f32 (the 128-channel levels, where 32-sample workgroups would leave half the CUs idle).
//
// What round 1 measured on its wide kernel (profiles/r01_*): the K loop issued MFMAs 79 % of the time and nearly all of
// the loss was the LDS *write* path — 3/4 of every K step's staging traffic was the weight slab, pushed through
// ds_write_b128 by the same four waves that issue the MFMAs.  Here the weights never touch LDS:
//   * at load time every conv's weights are repacked into MFMA B-FRAGMENT order (pack_fragments): for (output slab,
//     K group, tap slot) one contiguous 1 KiB block [lane 0..63][4] holding exactly the B operands of four consecutive
//     MFMAs.  A wave streams its blocks with one coalesced global_load_dwordx4 per block, one whole K step ahead of use.
//   * a wave = (output slab s, K slice ks): it accumulates ALL output tiles of its slab over its share of every K
//     chunk, so each B fragment is loaded by exactly one wave of the workgroup, every A fragment read from LDS feeds up
//     to 5 (+1) MFMA groups, all waves run the same instruction stream, and the K-slice partial tiles are summed when
//     the epilogue reads them back from LDS.
//   * only the activations go through LDS: [Lin][MS samples][KC + 4] per chunk in a ring of THREE stages, so chunk k+1
//     is already visible while chunk k is consumed: the first A fragment of the next chunk is read BEFORE the step's
//     barrier and no wave waits for LDS latency behind a barrier; one barrier per chunk.
//   * the step's global loads (weight fragments of chunk k+1, activation chunk k+3 -> staging registers) and the
//     ds_writes of chunk k+2 (fetched during the previous step) are spread over the step, one item per slot after an MFMA
//     round.
//   * with one wave per SIMD every VALU instruction between two MFMAs delays the next MFMA's issue (measured: none of the
//     K loop's loss was memory latency or the barrier), so the step contains no address arithmetic: loads take the
//     scalar-base + constant 32-bit vector-offset form, the K loop is unrolled over the 3 ring stages x 2 register sets
//     so that every LDS address is an invariant register + immediate, the workgroup mapping uses shifts and the kernel
//     arguments are fetched as one scalar batch.
// The including file includes this file only through unet.hip (single-unit builds) or kernel_shard.hip (sharded build).
// grid = Cout / CG channel groups x ceil(B / MS) sample tiles, dealt to the 8 XCDs so that the sum of weight and activation
// fetches over the 8 L2s is least (xcd_split).
#pragma once
#include <type_traits>

#include "params.h"

namespace edmp {

// native 16-byte vector for register staging (a float4 STRUCT copied global -> array -> LDS stays a memcpy through
// scratch memory: SROA only promotes arrays whose elements are loaded/stored as values)
using f32x4 = __attribute__((ext_vector_type(4))) float;

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [B, E).  All register arrays of the kernel are indexed
// with such constants: a runtime index, even one that would fold after unrolling, parks the array in scratch memory.
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// WK_K5K2: the Conv1dBlock at L = 2 in Karatsuba form.  With two positions the k5 conv is the 2x2 Toeplitz product
//     y0 = w2 x0 + w3 x1,  y1 = w1 x0 + w2 x1   (taps 0 and 4 only ever meet zero padding)
// = four (Cin x Cout) matrix products per sample block.  Three suffice: P = w2 (x0 + x1), Q = (w3 - w2) x1,
// R = (w1 - w2) x0, y0 = P + Q, y1 = P + R: the weight differences are formed once at load, x0 + x1 when the chunk is
// staged into LDS, P / Q / R accumulate in three tiles and are combined when the epilogue reads them back - 25 % fewer
// MFMAs on the twelve L = 2 convolutions of every forward, exact in exact arithmetic (fp32 rounding of the sums aside).
// WK_K5K4: the Conv1dBlock at L = 4 with the same idea applied twice.  y = T x with the banded 4x4 Toeplitz matrix T of
// the five taps; in 2x2 blocks T = [[A, B], [C, A]], so y_top = A (x_top + x_bot) + (B - A) x_bot, y_bot = A (x_top + x_bot)
// + (C - A) x_top: three 2x2 Toeplitz products, each done in the three-product form above: 9 matrix products instead of
// the 14 valid (tap, position) pairs of the direct form.  Nine staged positions (sums of input positions), nine weight
// slots (signed sums of taps, formed once at load), nine accumulator tiles, each output = the sum of four of them.
enum WideKind { WK_K5 = 0, WK_DOWN = 1, WK_UP = 2, WK_K5K2 = 3, WK_K5K4 = 4 };

// MS: samples per workgroup = MFMA tile height (32 or 16); CG: output channels per workgroup; GS: channels per GroupNorm
// group (WK_K5; CG % GS == 0); LIN: input positions; RES: fold the block's residual 1x1 conv (WK_K5)
template <int KIND, int MS, int CG, int GS, int LIN, bool RES>
struct WideCfg {
    static constexpr int NW = 4;                     // waves per workgroup
    static constexpr bool K2 = (KIND == WK_K5K2), K4 = (KIND == WK_K5K4);
    static constexpr bool BIL = K2 || K4;            // bilinear (Karatsuba) form: staged positions / accumulators are sums
    static constexpr int LLOAD = LIN;                // input positions fetched from HBM per chunk
    static constexpr int L = K2 ? 3 : K4 ? 9 : LIN;  // positions staged in LDS
    static constexpr int LOUT = (KIND == WK_K5 || BIL) ? LIN : (KIND == WK_DOWN) ? (LIN - 1) / 2 + 1 : ((2 * LIN == 8 || 2 * LIN == 14 || 2 * LIN == 26) ? 2 * LIN - 1 : 2 * LIN);
    static constexpr int LACC = K2 ? 3 : K4 ? 9 : LOUT;  // accumulator tiles per slab
    static constexpr bool GN = (KIND == WK_K5 || BIL);   // GroupNorm + Mish + add epilogue (else: + bias)
    // bilinear tables.  Staged position v = sum of the input positions lp with vcoef(v, lp); accumulator a collects staged
    // position a x weight slot a; output l = sum of the accumulators a with ocoef(l, a); rawpos(v) = lp if v IS x_lp.
    static constexpr bool vcoef(int v, int lp) {
        if (K2) return v == 2 || v == lp;
        if (K4) {
            constexpr int m[9] = {0xF, 0xA, 0x5, 0xC, 0x8, 0x4, 0x3, 0x2, 0x1};  // bit lp set: x_lp is part of the sum
            return (m[v] >> lp) & 1;
        }
        return v == lp;
    }
    static constexpr bool ocoef(int l, int a) {
        if (K2) return a == 0 || a == 1 + l;
        if (K4) {
            constexpr int m[4] = {0x01B, 0x02D, 0x0C3, 0x145};  // y0 = a0+a1+a3+a4, y1 = a0+a2+a3+a5, y2 = a0+a1+a6+a7, y3 = a0+a2+a6+a8
            return (m[l] >> a) & 1;
        }
        return l == a;
    }
    static constexpr int rawpos(int v) {
        if (K2) return v < 2 ? v : -1;
        if (K4) return v == 8 ? 0 : v == 7 ? 1 : v == 5 ? 2 : v == 4 ? 3 : -1;
        return v;
    }
    static constexpr int SW = MS;                    // output channels per slab (MFMA tile width = height)
    static constexpr int KG = (MS == 32) ? 8 : 16;   // channels per K group = four MFMAs (K = 2 resp. 4 each)
    static constexpr int AR = (MS == 32) ? 16 : 4;   // accumulator registers per tile
    static constexpr int S = CG / SW;                // output slabs per workgroup
    static constexpr int KSPLIT = NW / S;            // waves sharing a slab, each with its own K slice
    // channels per staged chunk = one barrier (64-channel chunks were measured neutral for the Karatsuba form: -2 % K loop, +1 k cycles of prologue)
    static constexpr int KC = (BIL && MS == 16) ? 64 : (KG * KSPLIT > 32) ? KG * KSPLIT : 32;  // (16-sample bilinear tiles: 64 channels make the staging map one item per thread)
    static constexpr int QW = KC / KG / KSPLIT;      // K groups per wave per chunk
    static constexpr int LDK = KC + 4;
    static constexpr int KT0 = (KIND == WK_K5 && LIN == 2) ? 1 : 0;  // first tap that can be valid
    static constexpr int NTAP = K2 ? 3 : K4 ? 9 : (KIND == WK_K5) ? ((LIN == 2) ? 3 : 5) : (KIND == WK_DOWN) ? 3 : 4;  // weight slots per K group
    static constexpr int NSLAB = NTAP + (RES ? 1 : 0);
    // weight slot of the pair (output tile l, input position lp), -1 if the tap does not exist
    static constexpr int slot(int l, int lp) {
        if (K2) return (l == 0 && lp == 2) ? 0 : (l == 1 && lp == 1) ? 1 : (l == 2 && lp == 0) ? 2 : -1;  // P, Q, R
        if (K4) return l == lp ? l : -1;
        const int t = (KIND == WK_K5) ? lp - l + 2 - KT0 : (KIND == WK_DOWN) ? lp - 2 * l + 1 : l + 1 - 2 * lp;
        return (t >= 0 && t < NTAP) ? t : -1;
    }
    static constexpr int A_FL = L * MS * LDK;        // floats per activation stage
    static constexpr int NTH = NW * 64;
    static constexpr int A_F4 = LLOAD * MS * (KC / 4);  // float4 items fetched per chunk
    static constexpr int NA = (A_F4 + NTH - 1) / NTH;
    static constexpr int NCOMMIT = BIL ? L : NA;        // ds_writes per thread per chunk (bilinear: one per STAGED position)
    static constexpr int NBL = QW * NSLAB;           // weight-fragment loads per wave per chunk
    static constexpr int YS = LOUT * CG + 4;  // (bilinear forms combine their product tiles into output tiles in registers before the spill)
    static constexpr int NP = KSPLIT;                // partial tiles per output element
    static constexpr int PPR = NTH / MS;             // threads per sample row in the final pass
    static constexpr int ROW_F4 = LOUT * CG / 4;     // float4 per sample row
    static constexpr int NF4 = (ROW_F4 + PPR - 1) / PPR;
    // MFMA blocks: one per (K group q, input position lp) = all tiles fed by that A fragment (+ the residual tile)
    static constexpr int NBLK = QW * L;
    static constexpr int NQSLOT = 4 * L * QW - L;    // bilinear K step: memory-work slots (one per (round, position), none in the last round)
    static constexpr int NSIDE = NA + NBL + NCOMMIT;  // side work items of a step: activation loads, weight loads, commits
    static constexpr int pairs() {
        int n = 0;
        for (int lp = 0; lp < L; ++lp)
            for (int l = 0; l < LACC; ++l)
                if (slot(l, lp) >= 0) ++n;
        return n;
    }
    static constexpr long valid_pairs() { return pairs(); }
    static constexpr size_t lds_bytes() {
        size_t a = 3 * (size_t)A_FL * sizeof(float);
        size_t y = (size_t)NP * MS * (size_t)YS * sizeof(float);
        return a > y ? a : y;
    }
    static_assert(MS == 32 || MS == 16, "tile height 32 (32x32x2 MFMA) or 16 (16x16x4 MFMA)");
    static_assert(NBLK >= 3 && 4 * (NBLK - 1) >= 1, "the memory work of a step is spread over the blocks before the last");
    static_assert(CG % SW == 0 && NW % S == 0 && S <= NW, "slabs per workgroup must divide the wave count");
    static_assert(!RES || KIND == WK_K5 || BIL, "the folded residual 1x1 conv belongs to a Conv1dBlock");
    static_assert(!BIL || (MS * (KC / 4) == NTH && (K2 ? LIN == 2 : LIN == 4)), "bilinear forms: staging item k of a thread = input position k of one (row, channel quad)");
    static_assert((LOUT * CG) % 4 == 0, "whole float4 columns");
    static_assert(!GN || CG % GS == 0, "whole GroupNorm groups per workgroup");
    static_assert(!GN || GS == CG || (4 * PPR) % CG == 0, "a thread's columns of the final pass lie in one GroupNorm group");
};

template <int MS>
struct WideAcc {
    using type = __attribute__((ext_vector_type(16))) float;
};
template <>
struct WideAcc<16> {
    using type = __attribute__((ext_vector_type(4))) float;
};

// The work of one workgroup: tile (channel group `grp`, sample tile `tile`) of the conv `p`; `lds` = the workgroup's dynamic LDS.
template <int KIND, int MS, int CG, int GS, int LIN, bool RES>
__device__ __forceinline__ void wide_conv_body(const RcbP& p, const int grp, const int tile, float* lds) {
    using Cf = WideCfg<KIND, MS, CG, GS, LIN, RES>;
    using acc_t = typename WideAcc<MS>::type;
    constexpr int L = Cf::L, LLOAD = Cf::LLOAD, LOUT = Cf::LOUT, LACC = Cf::LACC, SW = Cf::SW, KG = Cf::KG, AR = Cf::AR;
    constexpr bool BIL = Cf::BIL;
    constexpr int NCOMMIT = Cf::NCOMMIT;
    constexpr int S = Cf::S, KC = Cf::KC, QW = Cf::QW, LDK = Cf::LDK, NTAP = Cf::NTAP, NSLAB = Cf::NSLAB;
    constexpr int A_FL = Cf::A_FL, NTH = Cf::NTH, A_F4 = Cf::A_F4, NA = Cf::NA, YS = Cf::YS, NP = Cf::NP;
    constexpr int NBLK = Cf::NBLK, NSIDE = Cf::NSIDE, NBL = Cf::NBL;

    EDMP_STAMP(0, 0)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = wave % S, ks = wave / S;
    const int co0 = grp * CG;
    const int b0 = tile * MS;
    const int ch1 = p.C1 / KC, ch2 = p.C2 / KC;
    const int nK = ch1 + ch2;
    const int NKG = (p.C1 + p.C2) / KG;

    // ---- activation staging map (chunk invariant): item e = tid + k*NTH -> (position, sample row, channel quad)
    // global byte offsets of the item's float4 within chunk 0 of source 1 / source 2 (32-bit, added to a uniform base:
    // the loads take the scalar-base + vector-offset form, no per-load address arithmetic), LDS float offset in a stage
    unsigned a_g1[NA];  // (a concatenated input has two halves of EQUAL width, launcher-checked: one offset serves both sources)
    int a_l[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int e = min(tid + k * NTH, A_F4 - 1);
        const int lp = e / (MS * (KC / 4)), rem = e % (MS * (KC / 4));
        const int row = rem / (KC / 4), c4 = (rem % (KC / 4)) * 4;
        const int sb = min(b0 + row, p.B - 1);
        a_g1[k] = 4u * (unsigned)(((sb - b0) * LLOAD + lp) * p.C1 + c4);  // relative to the workgroup's first sample
        a_l[k] = lp * (MS * LDK) + row * LDK + c4;
    }
    // ---- weight fragment stream of this wave
    // (uniform base; the lane's 16 bytes within a fragment are the 32-bit vector offset of the scalar-base load form)
    const float* wb = p.W + ((size_t)(grp * S + s) * NKG) * (NSLAB * 256);
    unsigned lane16 = 16u * lane;
    // EDMP_OPAQUE keeps a 32-bit offset from being widened outside the K loop, where instruction selection would no longer
    // see `uniform base + zext(offset)` and would fall back to 64-bit vector adds per load; EDMP_OPAQUE_S keeps a uniform
    // byte offset in a scalar register, to be added to the scalar base (the immediate offset of a load reaches 4095 bytes)
#define EDMP_OPAQUE(v) asm volatile("" : "+v"(v))
#define EDMP_OPAQUE_S(v) asm volatile("" : "+s"(v))
    auto refresh_offsets = [&]() __attribute__((always_inline)) {
        EDMP_OPAQUE(lane16);
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            EDMP_OPAQUE(a_g1[k]);
        }
    };
    // weight fragment j of a step: fragments come in runs of four per scalar base (4 x 1 KiB = the immediate range)
    auto load_frag = [&](const float* w, auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        unsigned run = (j / 4) * 4096u;  // byte offset of the run, kept out of the load's immediate
        if constexpr (j >= 4) EDMP_OPAQUE_S(run);
        return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(w) + run + lane16 + (j % 4) * 1024);
    };

    acc_t acc[LACC];
    acc_t racc[RES ? LLOAD : 1];
#pragma unroll
    for (int t = 0; t < LACC; ++t)
#pragma unroll
        for (int i = 0; i < AR; ++i) acc[t][i] = 0.0f;
    if constexpr (RES) {
#pragma unroll
        for (int t = 0; t < LLOAD; ++t)
#pragma unroll
            for (int i = 0; i < AR; ++i) racc[t][i] = 0.0f;
    }

    f32x4 raA[NA], raB[NA];  // activation chunks in flight: fetched during one K step, committed to LDS during the next (two steps ahead: measured neutral)
    float4 bA[QW][NSLAB], bB[QW][NSLAB];

    auto load_a = [&](int nc, f32x4(&r)[NA]) __attribute__((always_inline)) {
        const bool first = nc < ch1;
        const char* srcb = reinterpret_cast<const char*>((first ? p.src1 + (size_t)b0 * LLOAD * p.C1 : p.src2 + (size_t)b0 * LLOAD * p.C2) + (first ? nc : nc - ch1) * KC);
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            r[k] = *reinterpret_cast<const f32x4*>(srcb + a_g1[k]);
        }
    };
    auto commit_a = [&](float* st, const f32x4(&r)[NA]) __attribute__((always_inline)) {
        if constexpr (BIL) {
            // staged position v = sum of the fetched positions in vcoef(v, .): this thread holds all of them for its
            // (sample row, channel quad)
            static_for<0, L>([&](auto vc) __attribute__((always_inline)) {
                constexpr int v = decltype(vc)::value;
                // (two packed-fp32 adds per sum: every VALU instruction in the K step costs an MFMA issue slot)
                f32x2_t slo = {0.f, 0.f}, shi = {0.f, 0.f};
                bool have = false;
                static_for<0, LLOAD>([&](auto lc) __attribute__((always_inline)) {
                    constexpr int lp = decltype(lc)::value;
                    if constexpr (Cf::vcoef(v, lp)) {
                        const f32x2_t rlo = {r[lp].x, r[lp].y}, rhi = {r[lp].z, r[lp].w};
                        slo = have ? pk_add(slo, rlo) : rlo;
                        shi = have ? pk_add(shi, rhi) : rhi;
                        have = true;
                    }
                });
                *reinterpret_cast<f32x4*>(st + a_l[0] + v * (MS * LDK)) = f32x4{slo.x, slo.y, shi.x, shi.y};
            });
        } else {
#pragma unroll
            for (int k = 0; k < NA; ++k)
                if ((A_F4 % NTH == 0) || tid + k * NTH < A_F4) *reinterpret_cast<f32x4*>(st + a_l[k]) = r[k];
        }
    };
    auto load_b = [&](int nc, float4(&b)[QW][NSLAB]) __attribute__((always_inline)) {
        const float* w = wb + ((size_t)(nc * (KC / KG) + ks * QW)) * (NSLAB * 256);
        refresh_offsets();
        static_for<0, QW * NSLAB>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            b[j / NSLAB][j % NSLAB] = load_frag(w, jc);
        });
    };

    // ---- prologue: weights of chunk 0 in flight, activation chunk 0 staged; chunks 1 and 2 are fetched and committed by the
    //      first K step, next to the fetch of chunk 3 (waiting for them here cost 1-2.5 us of every launch at L >= 4)
    f32x4 r1[NA];  // activation chunk 1 (the first step's extra staging set)
    load_a(0, raB);  // first: its data is on the way to the first MFMA twice (commit, barrier, fragment read) ...
    load_b(0, bA);   // ... the weights only once, and memory returns in request order
    // epilogue operands requested now (they land long before they are used; their pointers are not among the preloaded
    // kernel arguments, so anything earlier would put a scalar-memory wait in front of the loads above)
    const float bias_v = (ks == 0) ? p.bias[co0 + s * SW + (lane & (SW - 1))] : 0.0f;
    float rbias_v = 0.0f;
    if constexpr (RES) rbias_v = (ks == 0) ? p.res_bias[co0 + s * SW + (lane & (SW - 1))] : 0.0f;
    commit_a(lds, raB);
    __syncthreads();
    EDMP_STAMP(0, 1)

    // A fragment of this lane: sample row lane % MS, channel quad lane / MS of the wave's K groups
    const int frag = (lane & (MS - 1)) * LDK + 4 * (lane / MS) + KG * QW * ks;
    float4 a4 = *reinterpret_cast<const float4*>(lds + frag);
    // bilinear forms: the fragments of every staged position of the current K group and (second set) of the next one,
    // read a whole K group ahead
    float4 av[2][BIL ? L : 1];
    if constexpr (BIL) {
#pragma unroll
        for (int v = 0; v < L; ++v) av[0][v] = *reinterpret_cast<const float4*>(lds + frag + v * (MS * LDK));
    }

    // one MFMA on component J of the current A fragment and of a weight fragment
#define EDMP_W_MFMA(ACC, B4, J)                                                                                   \
    if constexpr (MS == 32) ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.J, (B4).J, ACC, 0, 0, 0);              \
    else ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.J, (B4).J, ACC, 0, 0, 0);

    // one K step: MFMAs of chunk i on stage `st` with fragments bc; fetch activation chunk i+2 and the weight fragments
    // of chunk i+1 (into bn); commit the fetched activations into stage `stw`; the first A fragment of chunk i+1 is read
    // from `stn` before the barrier.
#ifdef EDMP_STAMPS
    long long bar_cyc = 0;
    const long long kl0 = clock64();
#endif
    // The stage ring position is a compile-time constant (the K loop below is unrolled over the 3 stages x 2 register
    // sets), so every LDS address of a step is an invariant base register + immediate offset: no address arithmetic
    // between the MFMAs.
    auto step = [&](auto first_c, auto stage_c, auto par_c, int i, float4(&bc)[QW][NSLAB], float4(&bn)[QW][NSLAB], f32x4(&rc)[NA], f32x4(&rl)[NA]) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value;  // step 0: also commits chunk 1 (r1) into `stn`
        constexpr int SG = decltype(stage_c)::value;      // stage holding chunk i; i+1 is in the next one, i+2 goes into the third
        const float* st = lds + SG * A_FL;
        float* stn = lds + ((SG + 1) % 3) * A_FL;
        float* stw = lds + ((SG + 2) % 3) * A_FL;
        constexpr int PAR = decltype(par_c)::value;       // bilinear forms: the av[] set that holds this step's first K group
        // fetch chunk i+3 into `rl`, commit chunk i+2 (fetched during the previous step, in `rc`) into `stw`
        const int nca = min(i + 3, nK - 1), ncb = min(i + 1, nK - 1);
        const bool first = nca < ch1;
        const char* srcb = reinterpret_cast<const char*>((first ? p.src1 + (size_t)b0 * LLOAD * p.C1 : p.src2 + (size_t)b0 * LLOAD * p.C2) + (first ? nca : nca - ch1) * KC);
        const float* w = wb + ((size_t)(ncb * (KC / KG) + ks * QW)) * (NSLAB * 256);
        // first step: sources of chunks 1 and 2
        const int nc1 = min(1, nK - 1), nc2 = min(2, nK - 1);
        const bool first1 = nc1 < ch1, first2 = nc2 < ch1;
        const char* srcb1 = reinterpret_cast<const char*>((first1 ? p.src1 + (size_t)b0 * LLOAD * p.C1 : p.src2 + (size_t)b0 * LLOAD * p.C2) + (first1 ? nc1 : nc1 - ch1) * KC);
        const char* srcb2 = reinterpret_cast<const char*>((first2 ? p.src1 + (size_t)b0 * LLOAD * p.C1 : p.src2 + (size_t)b0 * LLOAD * p.C2) + (first2 ? nc2 : nc2 - ch1) * KC);
        refresh_offsets();
        // side work: the step's memory items (NBL weight loads, NA activation loads, the commits of the chunk fetched one
        // step earlier, in that order; the first step also commits chunk 1) are spread over SLOTS: one slot after
        // each of the four component rounds of every MFMA block except the last block (whose shadow is too short for a
        // ds_write to complete before the step's barrier)
        auto side = [&](auto xc) __attribute__((always_inline)) {
            constexpr int X = decltype(xc)::value;  // slot index
            constexpr int NSLOT = BIL ? Cf::NQSLOT : 4 * (NBLK - 1);
            constexpr int NX = FIRST ? 2 * NA : 0;    // extra fetches of the first step: chunks 1 (r1) and 2 (rc)
            constexpr int NXC = FIRST ? NCOMMIT : 0;  // ... and its extra commits (chunk 1)
            constexpr int NITEM = NX + NBL + NA + NXC + NCOMMIT;
            // commit item k of a staged chunk: k < NA the fetched positions, k == NA (Karatsuba form) position 2 = x0 + x1
            auto commit_item = [&](auto kc, float* stp, const f32x4(&r)[NA]) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                if constexpr (BIL) {  // staged position k = sum of the fetched positions in vcoef(k, .)
                    f32x2_t slo = {0.f, 0.f}, shi = {0.f, 0.f};
                    bool have = false;
                    static_for<0, LLOAD>([&](auto lc) __attribute__((always_inline)) {
                        constexpr int lp = decltype(lc)::value;
                        if constexpr (Cf::vcoef(k, lp)) {
                            const f32x2_t rlo = {r[lp].x, r[lp].y}, rhi = {r[lp].z, r[lp].w};
                            slo = have ? pk_add(slo, rlo) : rlo;
                            shi = have ? pk_add(shi, rhi) : rhi;
                            have = true;
                        }
                    });
                    *reinterpret_cast<f32x4*>(stp + a_l[0] + k * (MS * LDK)) = f32x4{slo.x, slo.y, shi.x, shi.y};
                } else {
                    if ((A_F4 % NTH == 0) || tid + k * NTH < A_F4) *reinterpret_cast<f32x4*>(stp + a_l[k]) = r[k];
                }
            };
            static_for<0, NITEM>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                if constexpr (j * NSLOT / NITEM == X) {
                    // (first step: chunks 1 and 2,) the weights of the next step (needed soonest), the activation fetch,
                    // the commits
                    if constexpr (j < NX) {
                        constexpr int k = j % NA;
                        if constexpr (j < NA) r1[k] = *reinterpret_cast<const f32x4*>(srcb1 + a_g1[k]);
                        else rc[k] = *reinterpret_cast<const f32x4*>(srcb2 + a_g1[k]);
                    } else if constexpr (j < NX + NBL) bn[(j - NX) / NSLAB][(j - NX) % NSLAB] = load_frag(w, std::integral_constant<int, j - NX>{});
                    else if constexpr (j < NX + NBL + NA) {
                        rl[j - NX - NBL] = *reinterpret_cast<const f32x4*>(srcb + a_g1[j - NX - NBL]);
                    }
                    else if constexpr (j < NX + NBL + NA + NXC) commit_item(std::integral_constant<int, j - NX - NBL - NA>{}, stn, r1);
                    else commit_item(std::integral_constant<int, j - NX - NBL - NA - NXC>{}, stw, rc);
                }
            });
        };
        if constexpr (BIL) {
            // Bilinear forms have ONE accumulator per staged position, so a block per position would be a chain of four
            // dependent MFMAs with memory work after every one of them.  Instead one block per K group: all staged
            // positions' fragments live in av[], rounds over the four fragment components, positions inside a round (=
            // consecutive MFMAs on different accumulators), one memory-work slot after every (round, position); a
            // position's fragment of the NEXT K group is read into place right after its last use (round w).
            static_for<0, QW>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value;
                constexpr int P = (PAR + q) & 1;
                static_for<0, 4>([&](auto jc) __attribute__((always_inline)) {
                    constexpr int J = decltype(jc)::value;
                    static_for<0, L>([&](auto vc) __attribute__((always_inline)) {
                        constexpr int v = decltype(vc)::value;
                        const float ax = (J == 0) ? av[P][v].x : (J == 1) ? av[P][v].y : (J == 2) ? av[P][v].z : av[P][v].w;
                        static_for<0, LACC>([&](auto ac) __attribute__((always_inline)) {
                            constexpr int a = decltype(ac)::value;
                            if constexpr (Cf::slot(a, v) >= 0) {
                                const float4& bq = bc[q][Cf::slot(a, v) >= 0 ? Cf::slot(a, v) : 0];
                                const float bx = (J == 0) ? bq.x : (J == 1) ? bq.y : (J == 2) ? bq.z : bq.w;
                                if constexpr (MS == 32) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, bx, acc[a], 0, 0, 0);
                                else acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, bx, acc[a], 0, 0, 0);
                            }
                        });
                        if constexpr (RES && Cf::rawpos(v) >= 0) {
                            const float4& bq = bc[q][NTAP];
                            const float bx = (J == 0) ? bq.x : (J == 1) ? bq.y : (J == 2) ? bq.z : bq.w;
                            if constexpr (MS == 32) racc[Cf::rawpos(v) >= 0 ? Cf::rawpos(v) : 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, bx, racc[Cf::rawpos(v) >= 0 ? Cf::rawpos(v) : 0], 0, 0, 0);
                            else racc[Cf::rawpos(v) >= 0 ? Cf::rawpos(v) : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, bx, racc[Cf::rawpos(v) >= 0 ? Cf::rawpos(v) : 0], 0, 0, 0);
                        }
                        if constexpr (J == 0 && !(FIRST && q + 1 == QW)) {  // the next K group's fragment of this position
                            const float* nx = (q + 1 < QW) ? st + frag + KG * (q + 1) : stn + frag;
                            av[P ^ 1][v] = *reinterpret_cast<const float4*>(nx + v * (MS * LDK));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        constexpr int X = (q * 4 + J) * L + v;
                        if constexpr (X < Cf::NQSLOT) {
                            side(std::integral_constant<int, X>{});
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                });
            });
        } else {
        static_for<0, QW>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value;
                static_for<0, L>([&](auto lpc) __attribute__((always_inline)) {
                    constexpr int lp = decltype(lpc)::value;
                    constexpr int X = q * L + lp;  // block index
                    // next A fragment: (lp+1, q) | (0, q+1) | first fragment of the next chunk
                    // (the first step reads the latter after its barrier: chunk 1 is committed during that step)
                    const float* an_p = (lp + 1 < L) ? st + frag + (lp + 1) * (MS * LDK) + KG * q
                                        : (q + 1 < QW) ? st + frag + KG * (q + 1)
                                                       : (FIRST ? st : stn) + frag;
                    const float4 an = *reinterpret_cast<const float4*>(an_p);
                    // the block: component-major over the tiles this A fragment feeds, so that consecutive MFMAs go to
                    // different accumulators (the 16x16x4 MFMA has 40 cycles of dependent latency for 32 of issue); after
                    // each round one slot of memory work rides in the shadow of the round's last MFMA
    #define EDMP_W_COMP(J, R)                                                                                                   \
        static_for<0, LACC>([&](auto lc) __attribute__((always_inline)) {                                                     \
            constexpr int l = decltype(lc)::value;                                                                              \
            if constexpr (Cf::slot(l, lp) >= 0) { EDMP_W_MFMA(acc[l], bc[q][Cf::slot(l, lp) >= 0 ? Cf::slot(l, lp) : 0], J) }   \
        });                                                                                                                     \
        if constexpr (RES && lp < LLOAD) { EDMP_W_MFMA(racc[lp < LLOAD ? lp : 0], bc[q][NTAP], J) }                             \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
        if constexpr (X < NBLK - 1) {                                                                                           \
            side(std::integral_constant<int, 4 * X + R>{});                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                                  \
        }
                    EDMP_W_COMP(x, 0)
                    EDMP_W_COMP(y, 1)
                    EDMP_W_COMP(z, 2)
                    EDMP_W_COMP(w, 3)
    #undef EDMP_W_COMP
                    a4 = an;
                });
            });
        }
#ifdef EDMP_STAMPS
        const long long tb0 = clock64();
        __syncthreads();
        bar_cyc += clock64() - tb0;
#else
        __syncthreads();
#endif
        if constexpr (FIRST) {
            if constexpr (BIL) {
#pragma unroll
                for (int v = 0; v < L; ++v) av[(PAR + QW) & 1][v] = *reinterpret_cast<const float4*>(stn + frag + v * (MS * LDK));
            } else {
                a4 = *reinterpret_cast<const float4*>(stn + frag);
            }
        }
    };

    {
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, QW & 1>;  // odd steps start on the other av[] set when a step has an odd number of K groups
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        using S2 = std::integral_constant<int, 2>;
        const std::false_type F{};
        step(std::true_type{}, S0{}, P0{}, 0, bA, bB, raA, raB);
        // step i: stage i % 3; odd i on the second weight / activation register sets
        for (int i = 1;;) {
            if (i >= nK) break;
            step(F, S1{}, P1{}, i++, bB, bA, raB, raA);
            if (i >= nK) break;
            step(F, S2{}, P0{}, i++, bA, bB, raA, raB);
            if (i >= nK) break;
            step(F, S0{}, P1{}, i++, bB, bA, raB, raA);
            if (i >= nK) break;
            step(F, S1{}, P0{}, i++, bA, bB, raA, raB);
            if (i >= nK) break;
            step(F, S2{}, P1{}, i++, bB, bA, raB, raA);
            if (i >= nK) break;
            step(F, S0{}, P0{}, i++, bA, bB, raA, raB);
        }
    }
#undef EDMP_W_MFMA
#undef EDMP_OPAQUE
#undef EDMP_OPAQUE_S
#ifdef EDMP_STAMPS
    if (blockIdx.x == 0 && lane == 0) {
        g_stamps[5][wave] = bar_cyc;
        g_stamps[5][4 + wave] = clock64() - kl0;
    }
#endif
    EDMP_STAMP(0, 2)

    // ---- epilogue: K-slice partial tiles (+bias in slice 0) -> LDS; then per thread (sample row, column part) the
    //      partials are summed and, for a Conv1dBlock, the per-(sample, GroupNorm group) statistics are reduced over the
    //      threads of the row, normalise, Mish, add; float4 stores (the closing barrier of the last step freed the stages)
    float* Y = lds;  // [NP][MS][YS]
    constexpr int PPR = Cf::PPR, ROW_F4 = Cf::ROW_F4, NF4 = Cf::NF4;
    constexpr bool GN = Cf::GN;
    const int erow = tid / PPR, epart = tid % PPR;
    const int eb = min(b0 + erow, p.B - 1);
    float4 g4[GN ? NF4 : 1], be4[GN ? NF4 : 1], ad4[GN ? NF4 : 1];
    if constexpr (GN) {
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int f = min(epart + PPR * i, ROW_F4 - 1);
            const int col = 4 * f;
            const int l = col / CG, ch = co0 + col % CG;
            g4[i] = *reinterpret_cast<const float4*>(p.gamma + ch);
            be4[i] = *reinterpret_cast<const float4*>(p.beta + ch);
            ad4[i] = make_float4(0.f, 0.f, 0.f, 0.f);  // one addend per launch: conv1 the time bias, conv2 the residual
            if (p.add_res) ad4[i] = *reinterpret_cast<const float4*>(p.add_res + ((size_t)eb * LOUT + l) * p.Cout + ch);
            else if (p.add_tb) ad4[i] = *reinterpret_cast<const float4*>(p.add_tb + ch);
        }
    }
    // accumulator element r of a lane: 32x32 tile: row (r&3) + 8*(r>>2) + 4*(lane>>5), column lane&31;
    //                                  16x16 tile: row 4*(lane>>4) + r, column lane&15
    float* Yw = Y + ks * (MS * YS) + s * SW + (lane & (SW - 1));
    auto acc_row = [&](int r) __attribute__((always_inline)) { return (MS == 32) ? (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) : 4 * (lane >> 4) + r; };
    if constexpr (RES) {
#pragma unroll
        for (int l = 0; l < LLOAD; ++l)
#pragma unroll
            for (int r = 0; r < AR; ++r) Yw[acc_row(r) * YS + l * CG] = racc[l][r] + rbias_v;
        __syncthreads();
        if (b0 + erow < p.B) {
#pragma unroll
            for (int i = 0; i < NF4; ++i) {
                const int f = epart + PPR * i;
                if ((ROW_F4 % PPR == 0) || f < ROW_F4) {
                    const int col = 4 * f;
                    const int l = col / CG, ch = co0 + col % CG;
                    float4 rv = *reinterpret_cast<const float4*>(Y + erow * YS + col);
#pragma unroll
                    for (int q = 1; q < NP; ++q) {
                        const float4 pv = *reinterpret_cast<const float4*>(Y + q * (MS * YS) + erow * YS + col);
                        rv.x += pv.x, rv.y += pv.y, rv.z += pv.z, rv.w += pv.w;
                    }
                    *reinterpret_cast<float4*>(p.res_out + ((size_t)(b0 + erow) * LLOAD + l) * p.Cout + ch) = rv;
                }
            }
        }
        __syncthreads();
    }
    // bilinear forms: output position l = sum of the product tiles in ocoef(l, .) - all in this wave's registers, in the
    // same lane layout, so the combination costs a few VALU adds here instead of LDS traffic for every product tile
    static_for<0, LOUT>([&](auto lc) __attribute__((always_inline)) {
        constexpr int l = decltype(lc)::value;
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            float y = bias_v;
            static_for<0, LACC>([&](auto ac) __attribute__((always_inline)) {
                constexpr int a = decltype(ac)::value;
                if constexpr (Cf::ocoef(l, a)) y += acc[a][r];
            });
            Yw[acc_row(r) * YS + l * CG] = y;
        }
    });
    __syncthreads();
    EDMP_STAMP(0, 3)
    {
        const int b = b0 + erow;
        float4 v[NF4];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int f = epart + PPR * i;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((ROW_F4 % PPR == 0) || f < ROW_F4) {
                {
                    v[i] = *reinterpret_cast<const float4*>(Y + erow * YS + 4 * f);
#pragma unroll
                    for (int q = 1; q < NP; ++q) {
                        const float4 pv = *reinterpret_cast<const float4*>(Y + q * (MS * YS) + erow * YS + 4 * f);
                        v[i].x += pv.x, v[i].y += pv.y, v[i].z += pv.z, v[i].w += pv.w;
                    }
                }
                sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
        }
        if constexpr (GN) {
            // the threads of a sample row that share this thread's GroupNorm group: all PPR of them when the workgroup
            // holds one group, else those whose column offset (4 * part) % CG falls into the same GS-wide group
            auto row_group_sum = [&](float x) __attribute__((always_inline)) {
                static_for<0, 4>([&](auto bc) __attribute__((always_inline)) {
                    constexpr int m = 1 << decltype(bc)::value;
                    if constexpr (m < PPR && (GS == CG || (4 * m) % CG < GS)) x = dpp_xor_add<m>(x);
                });
                return x;
            };
            constexpr float inv_n = 1.0f / (float)(LOUT * GS);
            const float mean = row_group_sum(sum) * inv_n;
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < NF4; ++i) {
                const int f = epart + PPR * i;
                if ((ROW_F4 % PPR == 0) || f < ROW_F4) {
                    const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
                    sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
                }
            }
            const float rstd = 1.0f / sqrtf(row_group_sum(sq) * inv_n + 1e-5f);
            if (b < p.B) {
#pragma unroll
                for (int i = 0; i < NF4; ++i) {
                    const int f = epart + PPR * i;
                    if ((ROW_F4 % PPR == 0) || f < ROW_F4) {
                        const int col = 4 * f;
                        const int l = col / CG, ch = co0 + col % CG;
                        const float s0 = rstd * g4[i].x, s1 = rstd * g4[i].y, s2 = rstd * g4[i].z, s3 = rstd * g4[i].w;
                        float4 o;
                        // (packed fp32: two elements per instruction for everything but min / exp2 / rcp)
                        const f32x2_t sa = {s0, s1}, sb = {s2, s3};
                        const f32x2_t ya = mish_fast2(f32x2_t{v[i].x, v[i].y} * sa + (f32x2_t{be4[i].x, be4[i].y} - sa * mean)) + f32x2_t{ad4[i].x, ad4[i].y};
                        const f32x2_t yb = mish_fast2(f32x2_t{v[i].z, v[i].w} * sb + (f32x2_t{be4[i].z, be4[i].w} - sb * mean)) + f32x2_t{ad4[i].z, ad4[i].w};
                        o.x = ya.x, o.y = ya.y, o.z = yb.x, o.w = yb.y;
                        *reinterpret_cast<float4*>(p.dst + ((size_t)b * LOUT + l) * p.Cout + ch) = o;
                    }
                }
            }
        } else if (b < p.B) {
#pragma unroll
            for (int i = 0; i < NF4; ++i) {
                const int f = epart + PPR * i;
                if ((ROW_F4 % PPR == 0) || f < ROW_F4) {
                    const int col = 4 * f;
                    const int l = col / CG, ch = co0 + col % CG;
                    *reinterpret_cast<float4*>(p.dst + ((size_t)b * LOUT + l) * p.Cout + ch) = v[i];
                }
            }
        }
    }
    EDMP_STAMP(0, 4)
}

// weight-fragment stream of one conv in HBM: [Cout/SW][Cin/KG][NSLAB][64 lanes][4] floats (pack_fragments)
// The parameters every workgroup needs before it can request its first bytes are leading scalar kernel arguments: with
// -mllvm -amdgpu-kernarg-preload-count=12 the command processor delivers them in SGPRs at wave launch, no scalar-memory round
// trip in front of the first loads (the struct `pr` carries the rest; its copies of these fields are ignored)
template <int KIND, int MS, int CG, int GS, int LIN, bool RES>
__global__ __launch_bounds__(256) void wide_conv_kernel(const float* a_src1, const float* a_src2, const float* a_W, int a_C1, int a_C2, int a_Cout, int a_B,
                                                        int a_gx_shift, int a_ng_shift, RcbP pr) {
    RcbP p = pr;
    p.src1 = a_src1, p.src2 = a_src2, p.W = a_W, p.C1 = a_C1, p.C2 = a_C2, p.Cout = a_Cout, p.B = a_B, p.gx_shift = a_gx_shift, p.ng_shift = a_ng_shift;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // workgroup -> (channel group, sample tile).  Consecutive workgroup ids go round-robin over the 8 XCDs (each with its own
    // L2): gx = 2^gx_shift of the XCDs split the channel groups, 8 / gx split the sample tiles, so a weight stream is fetched from HBM by
    // 8 / gx L2s and an activation tile by gx of them - the host picks the split with the least traffic (unet.hip: xcd_split)
    int grp, tile;
    {
        // (shifts, not divisions: a runtime integer division is ~40 instructions in front of the first load)
        const int lin = blockIdx.x;
        if (p.gx_shift >= 0) {  // ng and gx are powers of two
            const int xcd = lin & 7, j = lin >> 3, ngp_shift = p.ng_shift - p.gx_shift;
            grp = ((j & ((1 << ngp_shift) - 1)) << p.gx_shift) + (xcd & ((1 << p.gx_shift) - 1));
            tile = ((j >> ngp_shift) << (3 - p.gx_shift)) + (xcd >> p.gx_shift);
        } else {
            const int ng = p.Cout / CG;
            grp = lin % ng;
            tile = lin / ng;
        }
    }
    wide_conv_body<KIND, MS, CG, GS, LIN, RES>(p, grp, tile, lds);
}

// host: [tap][Cout][Cin] (taps 0..4; tap index 5 = the folded residual 1x1 conv) -> fragment stream
// [Cout/sw][Cin/kg][nslab][64][4]; slot t is tap kt0 + t for t < ntap and the residual slab for t == ntap.
// sw = 32: lane (n = lane%32, kh = lane/32) holds W[tap][slab*32 + n][8*kg + 4*kh + 0..3]   (four 32x32x2 MFMAs)
// sw = 16: lane (n = lane%16, kq = lane/16) holds W[tap][slab*16 + n][16*kg + 4*kq + 0..3]  (four 16x16x4 MFMAs)
inline void pack_fragments(const float* w_tco_ci, int cout, int cin, int kt0, int ntap, bool res, float* out, int sw = 32) {
    const int nslab = ntap + (res ? 1 : 0);
    const int kgc = (sw == 32) ? 8 : 16;
    const int nkg = cin / kgc;
    for (int sl = 0; sl < cout / sw; ++sl)
        for (int kg = 0; kg < nkg; ++kg)
            for (int t = 0; t < nslab; ++t) {
                const int tap = (t < ntap) ? kt0 + t : 5;
                float* o = out + (((size_t)sl * nkg + kg) * nslab + t) * 256;
                for (int lane = 0; lane < 64; ++lane) {
                    const int n = lane % sw, kh = lane / sw;
                    const float* src = w_tco_ci + ((size_t)tap * cout + sl * sw + n) * cin + kgc * kg + 4 * kh;
                    for (int j = 0; j < 4; ++j) o[lane * 4 + j] = src[j];
                }
            }
}

// Karatsuba form of the L = 2 Conv1dBlock (WK_K5K2): [tap][Cout][Cin] taps 1..3 (+ residual at tap index 5) ->
// slot 0 = w2, slot 1 = w3 - w2, slot 2 = w1 - w2 (+ the residual slab), then the fragment stream of pack_fragments
inline void pack_fragments_k2(const float* w_tco_ci, int cout, int cin, bool res, float* out) {
    std::vector<float> t((size_t)6 * cout * cin, 0.0f);
    const size_t n = (size_t)cout * cin;
    for (size_t i = 0; i < n; ++i) {
        const float w1 = w_tco_ci[1 * n + i], w2 = w_tco_ci[2 * n + i], w3 = w_tco_ci[3 * n + i];
        t[0 * n + i] = w2;
        t[1 * n + i] = w3 - w2;
        t[2 * n + i] = w1 - w2;
        if (res) t[5 * n + i] = w_tco_ci[5 * n + i];
    }
    pack_fragments(t.data(), cout, cin, 0, 3, res, out, 32);
}

// nested Karatsuba form of the L = 4 Conv1dBlock (WK_K5K4): nine weight slots = signed sums of the five taps
// (rows: slot, columns: tap 0..4), formed in double and rounded once; then the fragment stream of pack_fragments
inline void pack_fragments_k4(const float* w_tco_ci, int cout, int cin, bool res, float* out) {
    static const int wc[9][5] = {{0, 0, 1, 0, 0},   {0, 0, -1, 1, 0},  {0, 1, -1, 0, 0},  {0, 0, -1, 0, 1}, {0, 0, 1, -1, -1},
                                 {0, -1, 1, 1, -1}, {1, 0, -1, 0, 0},  {-1, 1, 1, -1, 0}, {-1, -1, 1, 0, 0}};
    const int nslot = 9 + (res ? 1 : 0);
    const size_t n = (size_t)cout * cin;
    std::vector<float> t((size_t)nslot * n, 0.0f);
    for (size_t i = 0; i < n; ++i) {
        for (int sl = 0; sl < 9; ++sl) {
            double a = 0.0;
            for (int k = 0; k < 5; ++k) a += wc[sl][k] * (double)w_tco_ci[(size_t)k * n + i];
            t[(size_t)sl * n + i] = (float)a;
        }
        if (res) t[(size_t)9 * n + i] = w_tco_ci[(size_t)5 * n + i];
    }
    // pack_fragments reads slot t < ntap from tap index kt0 + t and the residual from tap index 5: lay the ten slabs out so
    const int nkg = cin / 8;
    for (int sl = 0; sl < cout / 32; ++sl)
        for (int kg = 0; kg < nkg; ++kg)
            for (int ts = 0; ts < nslot; ++ts) {
                float* o = out + (((size_t)sl * nkg + kg) * nslot + ts) * 256;
                for (int lane = 0; lane < 64; ++lane) {
                    const int nn = lane % 32, kh = lane / 32;
                    const float* src = t.data() + ((size_t)ts * cout + sl * 32 + nn) * cin + 8 * kg + 4 * kh;
                    for (int j = 0; j < 4; ++j) o[lane * 4 + j] = src[j];
                }
            }
}

// How the 8 XCDs split a (channel groups x sample tiles) grid: gx of them across the groups, 8 / gx across the tiles.
// HBM fetches = weights x (8 / gx) + activations x gx (each L2 fetches what its workgroups touch); 0 = grid does not divide.
inline int xcd_split(int ng, int nt, double w_elems, double a_elems) {
    int best = 0;
    double cost = 0.0;
    for (int gx = 1; gx <= 8; gx *= 2) {
        if (ng % gx || nt % (8 / gx)) continue;
        const double c = w_elems * (8 / gx) + a_elems * gx;
        if (!best || c < cost) best = gx, cost = c;
    }
    return best;
}

template <int KIND, int MS, int CG, int GS, int LIN, bool RES>
int launch_wide_t(const RcbP& p, hipStream_t s) {
    // set once per instance; launches come from several host threads (two contexts: scenes in flight, chains of one batch)
    static std::atomic<int> attr_set{0};
    constexpr size_t bytes = WideCfg<KIND, MS, CG, GS, LIN, RES>::lds_bytes();
    static_assert(bytes <= 160 * 1024, "position-tile conv kernel exceeds the 160 KiB LDS of a CU");
    if (!attr_set.load(std::memory_order_acquire)) {  // idempotent call: a second thread racing here at worst repeats it
        EDMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_conv_kernel<KIND, MS, CG, GS, LIN, RES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        attr_set.store(1, std::memory_order_release);
    }
    const int ng = p.Cout / CG, nt = (p.B + MS - 1) / MS;
    EDMP_REQUIRE(p.C2 == 0 || p.C2 == p.C1, "wide_conv_kernel: the two halves of a concatenated input must have the same width (C1=%d, C2=%d)", p.C1, p.C2);
    RcbP q = p;
    const int gx = xcd_split(ng, nt, (double)p.Cout * (p.C1 + p.C2) * WideCfg<KIND, MS, CG, GS, LIN, RES>::NSLAB, (double)nt * MS * WideCfg<KIND, MS, CG, GS, LIN, RES>::LLOAD * (p.C1 + p.C2));
    q.gx_shift = q.ng_shift = -1;
    if (gx > 0 && (ng & (ng - 1)) == 0) {
        q.gx_shift = __builtin_ctz(gx);
        q.ng_shift = __builtin_ctz(ng);
    }
    hipLaunchKernelGGL((wide_conv_kernel<KIND, MS, CG, GS, LIN, RES>), dim3(ng * nt), dim3(256), bytes, s, q.src1, q.src2, q.W, q.C1, q.C2, q.Cout, q.B, q.gx_shift, q.ng_shift, q);
    return EDMP_OK;
}

}  // namespace edmp
