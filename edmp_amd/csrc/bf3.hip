// bf3.hip — the position-tile convolution of wide.hip on the bf16 matrix pipe with EXACT products ("bf16x3 split", round 6).
//
// gfx950 runs fp32 MFMAs at 1/16 of the bf16 rate.  An fp32 number is exactly hi + mid + lo with three bf16 numbers (8 + 8 + 8
// significand bits, round-to-nearest at every step: |mid| <= 2^-9 |x|, |lo| <= 2^-18 |x|, every difference exact), and the product
// of two bf16 numbers is exact in fp32.  So a * w = the sum of nine exact partial products; the three smallest (mid*lo, lo*mid,
// lo*lo: <= 2^-26 |a w| together, random sign - below the 2^-24 rounding of fp32's own accumulation) are dropped and SIX
// v_mfma_f32_16x16x32_bf16 (K = 32, 16 cycles each) do the work of sixteen v_mfma_f32_16x16x4_f32 (32 cycles each): 96 instead of
// 512 matrix-pipe cycles per 16 x 16 x 32 block.  Accumulation is fp32, in TWO accumulators per output tile: `big` takes the
// hi*hi products only (one rounding at full magnitude per 32-term dot product), `sml` the five correction products (2^-9 of the
// magnitude, so their roundings are 2^-9 of fp32's); they are added once, in the epilogue.  Measured against a float64 evaluation
// the result is CLOSER than the fp32-MFMA kernel's (profiles/r06_bf16x3.md), i.e. this is fp32 arithmetic with a different -
// shorter - rounding chain, not reduced precision.  (Reference op: diffusion/models/blocks.py:22-28 Conv1dBlock, :147-164 the adds
// and the residual 1x1 conv of ResidualConvolutionBlock, :213 / :251 the resampling convs.)
//
// Same contract as wide_conv_kernel<KIND, MS, CG, GS, LIN, RES> (wide.hip): a workgroup owns MS samples x CG = 32 output channels
// (whole GroupNorm groups) x ALL output positions; an accumulator tile is (one output position, 16 samples) x (16 channels); the
// pair (output l, input lp) contributes A[lp] x W[tap(l, lp)] iff that tap exists.  What differs:
//   * eight waves: four issue MFMAs, four STAGE: global fp32 -> split into three bf16 planes (v_cvt_pk_bf16_f32: round to nearest)
//     -> LDS (two stages), two chunks ahead in registers.  The split's VALU work (4.5 instructions per value) runs on the same SIMDs
//     BESIDE the matrix pipe, in other waves' issue slots, not between a wave's own MFMAs.
//   * MFMA wave = (16-channel slab, part): MS = 32: part = a 16-sample half, all tiles; MS = 16: part = a set of output positions
//     with (nearly) equal numbers of (l, lp) pairs, found by exhaustive search at compile time (Bf3Cfg::part0_mask).
//   * a chunk = 32 channels = ONE MFMA K.  It is worked in two PHASES with equal MFMA counts - A: the weight components lo and mid
//     (products hi*lo, mid*mid, hi*mid), B: the weight component hi (lo*hi, mid*hi, hi*hi) - and a phase's weight registers are
//     refilled with the next chunk's right after the phase: every weight fragment is requested half a chunk before its use with
//     3 x NSLOT x 4 registers of weights in flight, no double buffer.
//   * weights: fragment stream of bf16 triples [Cout/16][Cin/32][slot][component][64 lanes][8 bf16] (1 KiB blocks, 1.5x the
//     bytes of the fp32 stream), split once at load (pack_fragments_bf3).
#pragma once
#include <type_traits>

#include "params.h"
#include "wide.hip"

namespace edmp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

// host: round-to-nearest-even bf16 of a finite float; x = c[0] + c[1] + c[2] exactly
inline unsigned short bf16_rne_bits(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
inline float bf16_bits_to_float(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
inline void split3_bf16_rne(float x, unsigned short (&c)[3]) {
    c[0] = bf16_rne_bits(x);
    const float r1 = x - bf16_bits_to_float(c[0]);
    c[1] = bf16_rne_bits(r1);
    const float r2 = r1 - bf16_bits_to_float(c[1]);
    c[2] = bf16_rne_bits(r2);
}

template <int KIND, int MS, int CG, int GS, int LIN, bool RES>
struct Bf3Cfg {
    static constexpr int NTH = 512, NMW = 4;  // threads; MFMA waves (waves 4..7 stage)
    static constexpr int LOUT = (KIND == WK_K5) ? LIN : (KIND == WK_DOWN) ? (LIN - 1) / 2 + 1 : ((2 * LIN == 8 || 2 * LIN == 14 || 2 * LIN == 26) ? 2 * LIN - 1 : 2 * LIN);
    static constexpr bool GN = (KIND == WK_K5);
    static constexpr int SW = 16, S = CG / SW, PARTS = NMW / S;  // CG = 32: two slabs x two parts; CG = 64: four slabs, one wave each
    static constexpr bool ROWSPLIT = (PARTS == 2 && MS == 32);  // the two waves of a slab: sample halves (MS = 32) | output-position sets (MS = 16)
    static constexpr int NR = (PARTS == 1) ? MS / 16 : 1;       // 16-sample row blocks per wave (CG = 64: both halves of the 32 samples, sharing every weight fragment)
    // the residual tiles in ONE accumulator where two would not fit the 256 registers (L = 13 with the folded residual conv: 26 tiles; the
    // residual 1x1 conv has K = Cin only, so its 6 roundings per 32-term dot product stay below the fp32-MFMA kernel's 8)
    static constexpr bool RES_ONE_ACC = RES && MS == 16 && LIN >= 13;
    static constexpr int KC = 32;                 // channels per chunk = K of v_mfma_f32_16x16x32_bf16
    static constexpr int RS = KC + 8;             // bf16 per staged row (80 B: the 16 rows of a ds_read_b128 phase hit all 64 banks once)
    static constexpr int PLANE = MS * RS;
    static constexpr int STAGE = LIN * 3 * PLANE;  // bf16 per stage: [position][component][row][RS]
    static constexpr int NTAP = (KIND == WK_K5) ? 5 : (KIND == WK_DOWN) ? 3 : 4;
    static constexpr int NSLOT = NTAP + (RES ? 1 : 0);
    static constexpr int WBLK = NSLOT * 3 * 1024;  // bytes of a (slab, K group) run of the weight stream
    static constexpr int slot(int l, int lp) {
        const int t = (KIND == WK_K5) ? lp - l + 2 : (KIND == WK_DOWN) ? lp - 2 * l + 1 : l + 1 - 2 * lp;
        return (t >= 0 && t < NTAP) ? t : -1;
    }
    // tiles 0..LOUT-1: conv outputs; LOUT..LOUT+LIN-1 (RES): the folded residual 1x1 conv, one tile per input position
    static constexpr int NTILE = LOUT + (RES ? LIN : 0);
    static constexpr int tslot(int tile, int lp) { return tile < LOUT ? slot(tile, lp) : (tile - LOUT == lp ? NTAP : -1); }
    static constexpr int pairs_of(int l) {  // MFMA groups per chunk of output position l (its residual tile rides with it)
        int n = (RES && l < LIN) ? 1 : 0;
        for (int lp = 0; lp < LIN; ++lp) n += slot(l, lp) >= 0 ? 1 : 0;
        return n;
    }
    static constexpr int total_pairs() {
        int n = 0;
        for (int l = 0; l < LOUT; ++l) n += pairs_of(l);
        return n;
    }
    // position-set split: the subset of output positions for part 0 whose pair count is closest to half (ties: fewer tiles)
    static constexpr unsigned part0_mask() {
        if (ROWSPLIT || PARTS == 1) return ~0u;
        const int tot = total_pairs();
        int pc[16] = {};
        for (int l = 0; l < LOUT; ++l) pc[l] = pairs_of(l);
        unsigned best = 0;
        int bestmax = 1 << 30;
        for (unsigned m = 1; m < (1u << LOUT) - 1; m += 2) {  // position 0 in part 0 (halves the search; the parts are interchangeable)
            int n = 0;
            for (int l = 0; l < LOUT; ++l) n += ((m >> l) & 1u) ? pc[l] : 0;
            const int mx = n > tot - n ? n : tot - n;
            if (mx < bestmax) bestmax = mx, best = m;
        }
        return best;
    }
    static constexpr unsigned P0 = part0_mask();
    static constexpr bool owned(int tile, int H) {
        if (ROWSPLIT || PARTS == 1) return true;
        const int l = tile < LOUT ? tile : tile - LOUT;
        return (int)((P0 >> l) & 1u) == (H == 0 ? 1 : 0);
    }
    static constexpr int local(int tile, int H) {
        int n = 0;
        for (int q = 0; q < tile; ++q) n += owned(q, H) ? 1 : 0;
        return n;
    }
    static constexpr int ntiles(int H) { return local(NTILE, H); }
    static constexpr int MAXT = ntiles(0) > ntiles(1) ? ntiles(0) : ntiles(1);
    static constexpr bool needed(int H, int lp) {  // does part H read input position lp?
        for (int t = 0; t < NTILE; ++t)
            if (owned(t, H) && tslot(t, lp) >= 0) return true;
        return false;
    }
    // the K step of a part is the sequence of entries q = phase * LIN + lp (phase 0, 1) over the needed positions; ordinal / successor
    // of an entry (an even number of entries per chunk: the two fragment register sets alternate consistently across chunks)
    static constexpr int ord(int H, int q) {
        int n = 0;
        for (int i = 0; i < q; ++i) n += needed(H, i % LIN) ? 1 : 0;
        return n;
    }
    static constexpr int next(int H, int q) {
        for (int i = q + 1; i < 2 * LIN; ++i)
            if (needed(H, i % LIN)) return i;
        return -1;
    }
    static constexpr int first(int H) { return needed(H, 0) ? 0 : next(H, 0); }
    // staging: float4 items [position][row][channel quad] of a chunk over the 256 staging threads
    static constexpr int A_F4 = LIN * MS * (KC / 4);
    static constexpr int NA = (A_F4 + 255) / 256;
    // epilogue
    static constexpr int YS = LOUT * CG + 4, RYS = LIN * CG + 4;
    static constexpr int PPR = NTH / MS, ROW_F4 = LOUT * CG / 4, NF4 = (ROW_F4 + PPR - 1) / PPR;
    static constexpr int RROW_F4 = LIN * CG / 4, RNF4 = (RROW_F4 + PPR - 1) / PPR;
    static constexpr size_t lds_bytes() {
        const size_t a = 2 * (size_t)STAGE * 2, y = (size_t)MS * (YS + (RES ? RYS : 0)) * 4;
        return a > y ? a : y;
    }
    static_assert((CG == 32 && S == 2 && PARTS == 2) || (CG == 64 && S == 4 && PARTS == 1 && MS == 32), "two 16-channel slabs x two parts, or four slabs = the four MFMA waves");
    static_assert(MS == 32 || MS == 16, "32 samples (sample-half parts) or 16 samples (position-set parts)");
    static_assert(!RES || KIND == WK_K5, "the folded residual 1x1 conv belongs to a Conv1dBlock");
    static_assert(!GN || CG % GS == 0, "whole GroupNorm groups per workgroup");
    static_assert((4 * PPR) % CG == 0, "a thread's columns of the final pass lie in one GroupNorm group");
    static_assert(LOUT <= 16, "part0_mask searches 2^LOUT subsets");
};

#ifdef EDMP_BF3_STAMPS
__device__ long long g_bf3_stamps[8][8];
#define EDMP_BF3_STAMP(i) \
    if (blockIdx.x == 0 && lane == 0) g_bf3_stamps[wave][i] = clock64();
#else
#define EDMP_BF3_STAMP(i)
#endif

template <int KIND, int MS, int CG, int GS, int LIN, bool RES>
__global__ __launch_bounds__(512) void bf3_conv_kernel(const float* a_src1, const float* a_src2, const void* a_W, int a_C1, int a_C2, int a_Cout, int a_B,
                                                       int a_gx_shift, int a_ng_shift, RcbP p) {
    using Cf = Bf3Cfg<KIND, MS, CG, GS, LIN, RES>;
    constexpr int LOUT = Cf::LOUT, KC = Cf::KC, RS = Cf::RS, PLANE = Cf::PLANE, STAGE = Cf::STAGE, NSLOT = Cf::NSLOT, NTILE = Cf::NTILE, MAXT = Cf::MAXT;
    constexpr int YS = Cf::YS, RYS = Cf::RYS, NA = Cf::NA, A_F4 = Cf::A_F4, WBLK = Cf::WBLK;
    constexpr bool GN = Cf::GN;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned short* stg = reinterpret_cast<unsigned short*>(lds_raw);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool mfma_wave = wave < 4;
    const int s = wave & (Cf::S - 1), part = (Cf::PARTS == 2) ? (wave >> 1) & 1 : 0;
    constexpr int NR = Cf::NR;
    int grp, tile;
    {
        const int lin = blockIdx.x;
        if (a_gx_shift >= 0) {  // as wide_conv_kernel: consecutive workgroup ids go round-robin over the 8 XCDs
            const int xcd = lin & 7, j = lin >> 3, ngp_shift = a_ng_shift - a_gx_shift;
            grp = ((j & ((1 << ngp_shift) - 1)) << a_gx_shift) + (xcd & ((1 << a_gx_shift) - 1));
            tile = ((j >> ngp_shift) << (3 - a_gx_shift)) + (xcd >> a_gx_shift);
        } else {
            const int ng = a_Cout / CG;
            grp = lin % ng;
            tile = lin / ng;
        }
    }
    const int co0 = grp * CG, b0 = tile * MS;
    const int ch1 = a_C1 / KC, ch2 = a_C2 / KC, nK = ch1 + ch2;
    EDMP_BF3_STAMP(0)
    // The workgroup claims its waves' WHOLE register budget (2 waves per SIMD x 256 VGPRs = the SIMD's file): no wave of another kernel
    // can share a CU with it.  Measured need, not tidiness: with the 124-244 registers the kernels actually use, waves of the guide's
    // gradient kernel (guide.hip: guide_kernel<GM_GRAD>) that landed on a CU beside a bf16x3 workgroup - row chains, two scenes in
    // flight: other streams - came back with lanes 48-63 of ONE register changed in ~0.1 % of the launches (one gradient element of one
    // row; never with the fp32-MFMA program, never with this line: profiles/r06_coresidency_fault.md).  Cause not found (no LDS
    // overrun: padding the allocation changes nothing; no register beyond the declared count in the ISA; a pure-VALU canary kernel
    // beside the same launches stays bit-exact); the single-stream path never shares a CU anyway (one workgroup per CU).
    asm volatile("v_mov_b32 v255, 0" ::: "v255");

    // ---- staging side (waves 4..7): item e = (position, sample row, channel quad) of a chunk
    const int ptid = tid & 255;
    unsigned a_g[NA];  // byte offset of the item within chunk 0 of its source, relative to the workgroup's first sample
    int a_l[NA];       // bf16 offset of the item's hi plane within a stage
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int e = min(ptid + k * 256, A_F4 - 1);
        const int lp = e / (MS * 8), rem = e % (MS * 8);
        const int row = rem >> 3, quad = rem & 7;
        const int sb = min(b0 + row, a_B - 1);
        a_g[k] = 4u * (unsigned)(((sb - b0) * LIN + lp) * a_C1 + 4 * quad);  // (a concatenated input has two halves of EQUAL width, launcher-checked)
        a_l[k] = lp * 3 * PLANE + row * RS + 4 * quad;
    }
    auto fetch = [&](int nc, f32x4(&x)[NA]) __attribute__((always_inline)) {
        const bool first = nc < ch1;
        const char* srcb = reinterpret_cast<const char*>((first ? a_src1 + (size_t)b0 * LIN * a_C1 : a_src2 + (size_t)b0 * LIN * a_C2) + (first ? nc : nc - ch1) * KC);
#pragma unroll
        for (int k = 0; k < NA; ++k) x[k] = *reinterpret_cast<const f32x4*>(srcb + a_g[k]);
    };
    auto commit = [&](unsigned short* st, const f32x4(&x)[NA]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            if ((A_F4 % 256 == 0) || ptid + k * 256 < A_F4) {
                // component m = bf16(remainder), remainder -= component (exact): hi, mid, lo
                f32x2_t lo2 = {x[k].x, x[k].y}, hi2 = {x[k].z, x[k].w};
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const bf16x2_t c01 = __builtin_convertvector(lo2, bf16x2_t), c23 = __builtin_convertvector(hi2, bf16x2_t);
                    *reinterpret_cast<u32x2_t*>(st + a_l[k] + m * PLANE) = u32x2_t{__builtin_bit_cast(unsigned, c01), __builtin_bit_cast(unsigned, c23)};
                    if (m < 2) {
                        lo2 = lo2 - __builtin_convertvector(c01, f32x2_t);
                        hi2 = hi2 - __builtin_convertvector(c23, f32x2_t);
                    }
                }
            }
        }
    };

    // ---- MFMA side (waves 0..3): weight stream of slab (grp, s): runs of WBLK bytes per K group, [slot][component][64 lanes][16 B]
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(a_W) + ((size_t)(grp * Cf::S + s) * nK) * WBLK + 16 * lane;
    u32x4_t bw[3][NSLOT];
    auto load_w = [&](int kg, auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        const unsigned char* w = wb + (size_t)kg * WBLK + m * 1024;
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) bw[m][j] = *reinterpret_cast<const u32x4_t*>(w + j * 3 * 1024);
    };
    f32x4_t big[MAXT][NR], sml[MAXT][NR];
#pragma unroll
    for (int a = 0; a < MAXT; ++a)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            big[a][r] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            sml[a][r] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    // A fragment of a lane: sample row lane % 16 of the wave's row block(s), channel octet lane / 16 (row block r: + 16 r rows)
    const int afrag = ((Cf::ROWSPLIT ? 16 * part : 0) + (lane & 15)) * RS + 8 * (lane >> 4);
    constexpr int RBLK = 16 * RS;

    // one chunk of part H on stage `st` in two PHASES: A = the weight components lo and mid, B = the weight component hi (87 of the
    // 174 MFMAs each at L = 7); inside a phase one entry per needed input position: its activation components against every tile
    // the position feeds, smallest products first, consecutive MFMAs on different accumulators (9-15 MFMAs = 144-240 cycles per
    // entry); the fragments of the NEXT entry are read one entry ahead (the first entry's right after the step's barrier; a ring of
    // three stages that lets them be read before it was measured: no gain, 1.7 k cycles more prologue - profiles/r06_bf16x3.md).
    // After phase A its weight components are refilled with the next chunk's, after phase B component hi: every weight fragment
    // is requested half a chunk (~1.4 k cycles) before its first use.
    auto chunk = [&](auto hc, const unsigned short* st, int kgn) __attribute__((always_inline)) {
        constexpr int H = decltype(hc)::value;
        u32x4_t av[2][NR][3];
        {   // first entry (phase A: components hi, mid)
            constexpr int lp0 = Cf::first(H) % LIN;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                av[0][r][0] = *reinterpret_cast<const u32x4_t*>(st + (lp0 * 3 + 0) * PLANE + r * RBLK + afrag);
                av[0][r][1] = *reinterpret_cast<const u32x4_t*>(st + (lp0 * 3 + 1) * PLANE + r * RBLK + afrag);
            }
        }
        static_for<0, 2>([&](auto pc) __attribute__((always_inline)) {
            constexpr int ph = decltype(pc)::value;
            static_for<0, LIN>([&](auto lpc) __attribute__((always_inline)) {
                constexpr int lp = decltype(lpc)::value;
                constexpr int q = ph * LIN + lp;
                if constexpr (Cf::needed(H, lp)) {
                    constexpr int buf = Cf::ord(H, q) & 1;
                    constexpr int qn = Cf::next(H, q);
                    if constexpr (qn >= 0) {  // next entry's fragments: phase A reads the components hi and mid, phase B all three
                        constexpr int lpn = qn % LIN, ncomp = (qn / LIN == 0) ? 2 : 3;
#pragma unroll
                        for (int r = 0; r < NR; ++r)
#pragma unroll
                            for (int m = 0; m < ncomp; ++m) av[buf ^ 1][r][m] = *reinterpret_cast<const u32x4_t*>(st + (lpn * 3 + m) * PLANE + r * RBLK + afrag);
                    }
                    // products (activation component ia, weight component wc), kept iff ia + wc <= 2; phase A: (hi, lo) (mid, mid) (hi, mid); phase B: (lo, hi) (mid, hi) (hi, hi)
                    static_for<0, 3>([&](auto kc) __attribute__((always_inline)) {
                        constexpr int k = decltype(kc)::value;
                        constexpr int ia = (ph == 0) ? (k == 1 ? 1 : 0) : 2 - k;
                        constexpr int wc = (ph == 0) ? (k == 0 ? 2 : 1) : 0;
                        static_for<0, NTILE>([&](auto tc) __attribute__((always_inline)) {
                            constexpr int tl = decltype(tc)::value;
                            if constexpr (Cf::owned(tl, H) && Cf::tslot(tl, lp) >= 0) {
                                constexpr int sl = Cf::tslot(tl, lp) >= 0 ? Cf::tslot(tl, lp) : 0;
                                constexpr int la = Cf::local(tl, H);
                                constexpr bool one_acc = Cf::RES_ONE_ACC && tl >= LOUT;
#pragma unroll
                                for (int r = 0; r < NR; ++r) {  // (the row blocks share the weight fragment)
                                    if constexpr ((ia == 0 && wc == 0) || one_acc)
                                        big[la][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av[buf][r][ia]), __builtin_bit_cast(bf16x8_t, bw[wc][sl]), big[la][r], 0, 0, 0);
                                    else
                                        sml[la][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av[buf][r][ia]), __builtin_bit_cast(bf16x8_t, bw[wc][sl]), sml[la][r], 0, 0, 0);
                                }
                            }
                        });
                    });
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            // refill with the next chunk's components (the last chunk re-reads its own: harmless)
            if constexpr (ph == 0) {
                load_w(kgn, std::integral_constant<int, 2>{});
                load_w(kgn, std::integral_constant<int, 1>{});
            } else {
                load_w(kgn, std::integral_constant<int, 0>{});
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---- prologue: chunk 0 staged, chunks 1 and 2 and the weights of chunk 0 in flight
    f32x4 xa[NA], xb[NA];
    if (!mfma_wave) {
        fetch(0, xa);
        fetch(min(1, nK - 1), xb);
        commit(stg, xa);
        fetch(min(2, nK - 1), xa);
    } else {
        load_w(0, std::integral_constant<int, 2>{});
        load_w(0, std::integral_constant<int, 1>{});
        load_w(0, std::integral_constant<int, 0>{});
    }
    __syncthreads();
    EDMP_BF3_STAMP(1)

    // one loop per role (the roles are wave-uniform; every wave passes the same nK barriers)
    auto mfma_loop = [&](auto hc) __attribute__((always_inline)) {
        for (int c = 0; c < nK; ++c) {
            chunk(hc, stg + (c & 1) * STAGE, min(c + 1, nK - 1));
            __syncthreads();
        }
    };
    if (mfma_wave) {
        if (Cf::ROWSPLIT || Cf::PARTS == 1 || part == 0) mfma_loop(std::integral_constant<int, 0>{});
        else mfma_loop(std::integral_constant<int, 1>{});
    } else {
        // step c: split + commit chunk c + 1 (fetched a step ago or in the prologue) into the other stage, fetch chunk c + 3 into its registers
        for (int c = 0; c < nK; c += 2) {
            if (c + 1 < nK) commit(stg + STAGE, xb);
            if (c + 3 < nK) fetch(c + 3, xb);
            __syncthreads();
            if (c + 1 < nK) {
                if (c + 2 < nK) commit(stg, xa);
                if (c + 4 < nK) fetch(c + 4, xa);
                __syncthreads();
            }
        }
    }
    EDMP_BF3_STAMP(2)

    // ---- epilogue: big + sml (+ bias) -> LDS [row][position * CG + channel] (conv tiles; the residual tiles behind them); then all
    //      512 threads: per sample row the GroupNorm(8) statistics by DPP / permlane over the row's threads, normalise, Mish,
    //      + time bias | residual, float4 stores (the closing barrier of the last step freed the stages)
    float* Y = reinterpret_cast<float*>(lds_raw);
    float* R = Y + MS * YS;
    constexpr int PPR = Cf::PPR, ROW_F4 = Cf::ROW_F4, NF4 = Cf::NF4;
    const int erow = tid / PPR, epart = tid % PPR;
    const int eb = min(b0 + erow, a_B - 1);
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
            if (p.add_res) ad4[i] = *reinterpret_cast<const float4*>(p.add_res + ((size_t)eb * LOUT + l) * a_Cout + ch);
            else if (p.add_tb) ad4[i] = *reinterpret_cast<const float4*>(p.add_tb + ch);
        }
    }
    auto spill = [&](auto hc) __attribute__((always_inline)) {
        constexpr int H = decltype(hc)::value;
        const int col = s * 16 + (lane & 15);
        const float bias_v = p.bias[co0 + col];
        float rbias_v = 0.f;
        if constexpr (RES) rbias_v = p.res_bias[co0 + col];
        const int row0 = (Cf::ROWSPLIT ? 16 * part : 0) + 4 * (lane >> 4);  // accumulator element r: row 4 * (lane / 16) + r, column lane % 16
        static_for<0, NTILE>([&](auto tc) __attribute__((always_inline)) {
            constexpr int tl = decltype(tc)::value;
            if constexpr (Cf::owned(tl, H)) {
                constexpr int la = Cf::local(tl, H);
#pragma unroll
                for (int rb = 0; rb < NR; ++rb) {
                    if constexpr (tl < LOUT) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) Y[(row0 + 16 * rb + r) * YS + tl * CG + col] = (big[la][rb][r] + sml[la][rb][r]) + bias_v;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) R[(row0 + 16 * rb + r) * RYS + (tl - LOUT) * CG + col] = (big[la][rb][r] + sml[la][rb][r]) + rbias_v;
                    }
                }
            }
        });
    };
    if (mfma_wave) {
        if (Cf::ROWSPLIT || Cf::PARTS == 1 || part == 0) spill(std::integral_constant<int, 0>{});
        else spill(std::integral_constant<int, 1>{});
    }
    __syncthreads();
    EDMP_BF3_STAMP(3)
    const int b = b0 + erow;
    if constexpr (RES) {  // the folded residual 1x1 conv: conv + res_bias -> res_out (conv2's residual addend)
        if (b < a_B) {
#pragma unroll
            for (int i = 0; i < Cf::RNF4; ++i) {
                const int f = epart + PPR * i;
                if ((Cf::RROW_F4 % PPR == 0) || f < Cf::RROW_F4) {
                    const int col = 4 * f;
                    const int l = col / CG, ch = co0 + col % CG;
                    *reinterpret_cast<float4*>(p.res_out + ((size_t)b * LIN + l) * a_Cout + ch) = *reinterpret_cast<const float4*>(R + erow * RYS + col);
                }
            }
        }
    }
    {
        float4 v[NF4];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int f = epart + PPR * i;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((ROW_F4 % PPR == 0) || f < ROW_F4) {
                v[i] = *reinterpret_cast<const float4*>(Y + erow * YS + 4 * f);
                sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
        }
        if constexpr (GN) {
            // the threads of a sample row that share this thread's GroupNorm group: lane ^ m for every m whose column step 4 m
            // stays inside the GS-wide group or is a multiple of CG
            auto row_group_sum = [&](float x) __attribute__((always_inline)) {
                static_for<0, 4>([&](auto bc) __attribute__((always_inline)) {
                    constexpr int m = 1 << decltype(bc)::value;
                    if constexpr (m < PPR && (GS == CG || (4 * m) % CG < GS)) x = dpp_xor_add<m>(x);
                });
                if constexpr (PPR > 16) x = swap16_add(x);  // (4 * 16) % CG == 0
                static_assert(PPR <= 32, "a sample row's threads lie in one 32-lane half of a wave");
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
            if (b < a_B) {
#pragma unroll
                for (int i = 0; i < NF4; ++i) {
                    const int f = epart + PPR * i;
                    if ((ROW_F4 % PPR == 0) || f < ROW_F4) {
                        const int col = 4 * f;
                        const int l = col / CG, ch = co0 + col % CG;
                        const float s0 = rstd * g4[i].x, s1 = rstd * g4[i].y, s2 = rstd * g4[i].z, s3 = rstd * g4[i].w;
                        const f32x2_t sa = {s0, s1}, sb = {s2, s3};
                        const f32x2_t ya = mish_fast2(f32x2_t{v[i].x, v[i].y} * sa + (f32x2_t{be4[i].x, be4[i].y} - sa * mean)) + f32x2_t{ad4[i].x, ad4[i].y};
                        const f32x2_t yb = mish_fast2(f32x2_t{v[i].z, v[i].w} * sb + (f32x2_t{be4[i].z, be4[i].w} - sb * mean)) + f32x2_t{ad4[i].z, ad4[i].w};
                        float4 o;
                        o.x = ya.x, o.y = ya.y, o.z = yb.x, o.w = yb.y;
                        *reinterpret_cast<float4*>(p.dst + ((size_t)b * LOUT + l) * a_Cout + ch) = o;
                    }
                }
            }
        } else if (b < a_B) {
#pragma unroll
            for (int i = 0; i < NF4; ++i) {
                const int f = epart + PPR * i;
                if ((ROW_F4 % PPR == 0) || f < ROW_F4) {
                    const int col = 4 * f;
                    const int l = col / CG, ch = co0 + col % CG;
                    *reinterpret_cast<float4*>(p.dst + ((size_t)b * LOUT + l) * a_Cout + ch) = v[i];
                }
            }
        }
    }
    EDMP_BF3_STAMP(4)
}

// host: [tap][Cout][Cin] (taps 0..ntap-1; tap index 5 = the folded residual 1x1 conv) -> bf16-triple fragment stream
// [Cout/16][Cin/32][nslot][component 3][64 lanes][8 bf16]: lane (n = lane % 16, oct = lane / 16) holds
// W_comp[slot][slab * 16 + n][32 kg + 8 oct + 0..7] - the B operand of v_mfma_f32_16x16x32_bf16.  `out` counts in bf16.
inline size_t bf3_stream_elems(int cout, int cin, int nslot) { return (size_t)(cout / 16) * (cin / 32) * nslot * 3 * 512; }
inline void pack_fragments_bf3(const float* w_tco_ci, int cout, int cin, int ntap, bool res, unsigned short* out) {
    const size_t n = (size_t)cout * cin;
    const int nkg = cin / 32, nslot = ntap + (res ? 1 : 0);
    for (int sl = 0; sl < cout / 16; ++sl)
        for (int kg = 0; kg < nkg; ++kg)
            for (int slot = 0; slot < nslot; ++slot) {
                const int tap = slot < ntap ? slot : 5;
                for (int lane = 0; lane < 64; ++lane) {
                    const int nn = lane % 16, oct = lane / 16;
                    for (int j = 0; j < 8; ++j) {
                        const float wv = w_tco_ci[(size_t)tap * n + (size_t)(sl * 16 + nn) * cin + 32 * kg + 8 * oct + j];
                        unsigned short c3[3];
                        split3_bf16_rne(wv, c3);
                        for (int comp = 0; comp < 3; ++comp) out[((((size_t)sl * nkg + kg) * nslot + slot) * 3 + comp) * 512 + lane * 8 + j] = c3[comp];
                    }
                }
            }
}

template <int KIND, int MS, int CG, int GS, int LIN, bool RES>
int launch_bf3_t(const RcbP& p, hipStream_t s) {
    using Cf = Bf3Cfg<KIND, MS, CG, GS, LIN, RES>;
    static std::atomic<int> attr_set{0};
    constexpr size_t bytes = Cf::lds_bytes();
    static_assert(bytes <= 160 * 1024, "bf16x3 position-tile kernel exceeds the 160 KiB LDS of a CU");
    if (!attr_set.load(std::memory_order_acquire)) {
        EDMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&bf3_conv_kernel<KIND, MS, CG, GS, LIN, RES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        attr_set.store(1, std::memory_order_release);
    }
    EDMP_REQUIRE(p.C1 % Cf::KC == 0 && p.C2 % Cf::KC == 0 && p.Cout % CG == 0, "bf3_conv_kernel: channels must be multiples of 32 (C1=%d, C2=%d, Cout=%d)", p.C1, p.C2, p.Cout);
    EDMP_REQUIRE(p.C2 == 0 || p.C2 == p.C1, "bf3_conv_kernel: the two halves of a concatenated input must have the same width (C1=%d, C2=%d)", p.C1, p.C2);
    const int ng = p.Cout / CG, nt = (p.B + MS - 1) / MS;
    const int gx = xcd_split(ng, nt, (double)p.Cout * (p.C1 + p.C2) * Cf::NSLOT * 1.5, (double)nt * MS * LIN * (p.C1 + p.C2));
    int gxs = -1, ngs = -1;
    if (gx > 0 && (ng & (ng - 1)) == 0) {
        gxs = __builtin_ctz(gx);
        ngs = __builtin_ctz(ng);
    }
    hipLaunchKernelGGL((bf3_conv_kernel<KIND, MS, CG, GS, LIN, RES>), dim3(ng * nt), dim3(512), bytes, s, p.src1, p.src2, reinterpret_cast<const void*>(p.W), p.C1, p.C2, p.Cout, p.B, gxs, ngs, p);
    return EDMP_OK;
}

}  // namespace edmp
