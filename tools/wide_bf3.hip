// wide_bf3.hip — EXPERIMENT (round 3; a tool, not part of libedmp_hip.so): the L = 2 Conv1dBlock of the 512-channel levels
// (wide.hip: WK_K5K2, the dominant instance of the layer program) on the bf16 matrix pipe with EXACT products.
//
// fp32 MFMA runs at 1/16 of the bf16 rate on gfx950.  An fp32 number is exactly the sum of three bf16 numbers
// (8 + 8 + 8 significand bits: hi = trunc16(x), mid = trunc16(x - hi), lo = x - hi - mid, every step exact), and the
// product of two bf16 numbers is exact in fp32 (16 significand bits).  So  a * b = sum over the 9 pairs (a_i, b_j)  holds
// exactly, and nine v_mfma_f32_32x32x16_bf16 with fp32 accumulation do the work of eight v_mfma_f32_32x32x2_f32 (K = 16)
// in 9 x 32 instead of 8 x 64 matrix-pipe cycles: 0.5625 of the time.  What differs from the fp32 instruction is only
// where the accumulation rounds (once per 16-term dot product of exact products instead of once per term) - the same
// class of difference as a change of summation order.  All nine terms are issued; no truncated (6- / 3-term) variant.
//
// Same decomposition as wide_conv_kernel<WK_K5K2, 32, 64, 64, 2, RES = false> (Karatsuba form: P = w2 (x0 + x1),
// Q = (w3 - w2) x1, R = (w1 - w2) x0; y0 = P + Q, y1 = P + R), same epilogue (K-slice partials -> LDS -> GroupNorm(8) ->
// Mish -> + time bias | residual).  Differences:
//   * weights: fragment stream of bf16 triples, [Cout/32][Cin/16][3 slots][3 components][64 lanes][8 bf16] - 1 KiB blocks
//     like the fp32 stream (1.5x its bytes), split once at load (pack_fragments_k2_bf3);
//   * activations: fetched as fp32, the staged sum x0 + x1 formed in fp32, every value split when the chunk is committed
//     to LDS as three bf16 planes;
//   * 8 waves: four issue MFMAs (output slab x K slice, as before), four stage the next chunk (global -> split -> LDS), so
//     the split's VALU work runs beside the matrix pipe on the same SIMDs instead of between its instructions.
#pragma once
#include "../edmp_amd/csrc/params.h"
#include "../edmp_amd/csrc/wide.hip"

namespace edmp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

// x = hi + mid + lo exactly, each the upper 16 bits of an fp32 (a bf16); returns the three bf16 bit patterns
__host__ __device__ __forceinline__ void split3_bf16(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
    const unsigned xb = __builtin_bit_cast(unsigned, x);
    const unsigned hb = xb & 0xffff0000u;
    const float r1 = x - __builtin_bit_cast(float, hb);
    const unsigned mb = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
    const float r2 = r1 - __builtin_bit_cast(float, mb);
    hi = hb >> 16;
    mid = mb >> 16;
    lo = __builtin_bit_cast(unsigned, r2) >> 16;
}

struct Bf3Cfg {
    static constexpr int MS = 32, CG = 64, GS = 64, LIN = 2, LOUT = 2;
    static constexpr int S = 2, KSPLIT = 2;        // output slabs per workgroup, K slices
    static constexpr int KC = 64;                  // channels per staged chunk
    static constexpr int KG = 16;                  // channels per MFMA (K of v_mfma_f32_32x32x16_bf16)
    static constexpr int QW = KC / KG / KSPLIT;    // K groups per MFMA wave per chunk
    static constexpr int RS = KC + 8;              // row stride of a staged plane, in bf16 (144 B: the 16 rows of a ds_read_b128 phase hit all 64 banks once)
    static constexpr int PLANE = MS * RS;          // bf16 per (position, component) plane
    static constexpr int STAGE = 9 * PLANE;        // bf16 per stage: [position v = x0, x1, x0 + x1][component][row][RS]
    static constexpr int YS = LOUT * CG + 4;
    static constexpr int NP = KSPLIT;
    static constexpr size_t lds_bytes() {
        const size_t a = 2 * (size_t)STAGE * 2, y = (size_t)NP * MS * YS * 4;
        return a > y ? a : y;
    }
};

#ifdef BF3_STAMPS
__device__ long long g_bf3_stamps[8][8];
__device__ unsigned long long g_bf3_wall[2] = {~0ull, 0ull};  // earliest workgroup start / latest workgroup end of the launch (100 MHz wall clock)
#define BF3_STAMP(i)                                                              \
    if (blockIdx.x == 0 && lane == 0) {                                           \
        g_bf3_stamps[wave][i] = clock64();                                        \
        if ((i) == 0 || (i) == 3) g_bf3_stamps[wave][(i) == 0 ? 6 : 7] = (long long)wall_clock64(); \
    }                                                                             \
    if ((i) == 0 && threadIdx.x == 0) atomicMin(&g_bf3_wall[0], wall_clock64()); \
    if ((i) == 3 && threadIdx.x == 0) atomicMax(&g_bf3_wall[1], wall_clock64());
#else
#define BF3_STAMP(i)
#endif
__global__ __launch_bounds__(512) void k2_bf3_kernel(const float* a_src1, const float* a_src2, const void* a_W, int a_C1, int a_C2, int a_Cout, int a_B,
                                                     int a_gx_shift, int a_ng_shift, RcbP p) {
    using Cf = Bf3Cfg;
    constexpr int MS = Cf::MS, CG = Cf::CG, GS = Cf::GS, LOUT = Cf::LOUT, KC = Cf::KC, KG = Cf::KG, QW = Cf::QW, RS = Cf::RS, PLANE = Cf::PLANE, STAGE = Cf::STAGE;
    constexpr int YS = Cf::YS, NP = Cf::NP;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned short* stg = reinterpret_cast<unsigned short*>(lds_raw);  // two stages of bf16 planes
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    const int s = wave & 1, ks = (wave >> 1) & 1;
    int grp, tile;
    {
        const int lin = blockIdx.x;
        if (a_gx_shift >= 0) {
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
    const int NKG = (a_C1 + a_C2) / KG;

    // ---- producer side: item = (sample row, channel quad) of the chunk; 32 rows x 16 quads = 512 items, two per producer thread
    const int ptid = tid - 256;
    auto fetch = [&](int nc, float4 (&x)[2][2]) __attribute__((always_inline)) {
        const bool first = nc < ch1;
        const float* src = first ? a_src1 : a_src2;
        const int Cs = first ? a_C1 : a_C2, c0 = (first ? nc : nc - ch1) * KC;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int e = ptid + 256 * it, row = e >> 4, q = e & 15;
            const int sb = min(b0 + row, a_B - 1);
#pragma unroll
            for (int lp = 0; lp < 2; ++lp) x[it][lp] = *reinterpret_cast<const float4*>(src + ((size_t)sb * 2 + lp) * Cs + c0 + 4 * q);
        }
    };
    auto commit = [&](unsigned short* st, const float4 (&x)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int e = ptid + 256 * it, row = e >> 4, q = e & 15;
            const float v0[4] = {x[it][0].x, x[it][0].y, x[it][0].z, x[it][0].w};
            const float v1[4] = {x[it][1].x, x[it][1].y, x[it][1].z, x[it][1].w};
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                // split four values: component = upper 16 bits of x, of x - hi, of x - hi - mid (each difference exact);
                // v_perm_b32 packs the upper halves of two registers in one instruction: 5.5 VALU per value
                unsigned short* base = st + (v * 3) * PLANE + row * RS + 4 * q;
                float x4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) x4[j] = v == 0 ? v0[j] : v == 1 ? v1[j] : v0[j] + v1[j];
#ifdef BF3_NOSPLIT
                *reinterpret_cast<u32x2_t*>(base) = u32x2_t{__builtin_bit_cast(unsigned, x4[0]), __builtin_bit_cast(unsigned, x4[1])};
                *reinterpret_cast<u32x2_t*>(base + PLANE) = u32x2_t{__builtin_bit_cast(unsigned, x4[2]), __builtin_bit_cast(unsigned, x4[3])};
                *reinterpret_cast<u32x2_t*>(base + 2 * PLANE) = u32x2_t{__builtin_bit_cast(unsigned, x4[2]), __builtin_bit_cast(unsigned, x4[1])};
#else
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    unsigned u[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) u[j] = __builtin_bit_cast(unsigned, x4[j]);
                    // upper halves of (u[1], u[0]) -> one dword (element 0 in the low half), likewise (u[3], u[2])
                    *reinterpret_cast<u32x2_t*>(base + m * PLANE) = u32x2_t{__builtin_amdgcn_perm(u[1], u[0], 0x07060302u), __builtin_amdgcn_perm(u[3], u[2], 0x07060302u)};
                    if (m < 2) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) x4[j] = x4[j] - __builtin_bit_cast(float, u[j] & 0xffff0000u);
                    }
                }
#endif
            }
        }
    };

    // ---- consumer side: weight stream of (slab, K slice); blocks of 1 KiB: [slot][component][64 lanes][16 B]
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(a_W) + ((size_t)(grp * Cf::S + s) * NKG) * (9 * 1024) + 16 * lane;
    u32x4_t bw[2][9];
    auto load_w = [&](int kg, u32x4_t (&b)[9]) __attribute__((always_inline)) {
        const unsigned char* w = wb + (size_t)kg * (9 * 1024);
#pragma unroll
        for (int j = 0; j < 9; ++j) b[j] = *reinterpret_cast<const u32x4_t*>(w + j * 1024);
    };
    f32x16_t acc[3];  // P (staged position 2 x slot 0), Q (position 1 x slot 1), R (position 0 x slot 2)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;

    // ---- prologue
    BF3_STAMP(0)
    float4 xr[2][2];
    if (!consumer) {
        fetch(0, xr);
        commit(stg, xr);
        if (nK > 1) fetch(1, xr);
    } else {
        load_w(ks * QW, bw[0]);
    }
    __syncthreads();
    BF3_STAMP(1)
#ifdef BF3_STAMPS
    long long work = 0;
#endif

    // A fragment of a lane: row lane % 32, channel octet lane / 32 of the K group
    const int afrag = (lane & 31) * RS + 8 * (lane >> 5);
    for (int c = 0; c < nK; ++c) {
        unsigned short* st = stg + (c & 1) * STAGE;
#ifdef BF3_STAMPS
        const long long tw0 = clock64();
#endif
        if (consumer) {
#pragma unroll
            for (int q = 0; q < QW; ++q) {
                const int kgl = ks * QW + q;  // K group within the chunk
                // this wave's next K group: the next of its share of the chunk, else the first of its share of the next chunk
                const int kgn = min((q + 1 < QW) ? c * (KC / KG) + kgl + 1 : (c + 1) * (KC / KG) + ks * QW, NKG - 1);
                load_w(kgn, bw[(q + 1) & 1]);
                u32x4_t av[3][3];
#pragma unroll
                for (int v = 0; v < 3; ++v)
#pragma unroll
                    for (int m = 0; m < 3; ++m) av[v][m] = *reinterpret_cast<const u32x4_t*>(st + (v * 3 + m) * PLANE + afrag + KG * kgl);
                const u32x4_t(&b)[9] = bw[q & 1];
                // nine exact partial products per accumulator, small terms first; consecutive MFMAs on different accumulators
                constexpr int order[9][2] = {{2, 2}, {2, 1}, {1, 2}, {2, 0}, {1, 1}, {0, 2}, {1, 0}, {0, 1}, {0, 0}};
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int ia = order[t][0], ib = order[t][1];
                    // accumulator a: staged position (2 - a), weight slot a
#pragma unroll
                    for (int a = 0; a < 3; ++a)
#ifdef BF3_NOMFMA
                        acc[a][t] += __builtin_bit_cast(float, av[2 - a][ia][0] ^ b[a * 3 + ib][1]);
#else
                        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av[2 - a][ia]), __builtin_bit_cast(bf16x8_t, b[a * 3 + ib]), acc[a], 0, 0, 0);
#endif
                }
            }
        } else if (c + 1 < nK) {
            commit(stg + ((c + 1) & 1) * STAGE, xr);
            if (c + 2 < nK) fetch(c + 2, xr);
        }
#ifdef BF3_STAMPS
        work += clock64() - tw0;
#endif
        __syncthreads();
    }
    BF3_STAMP(2)
#ifdef BF3_STAMPS
    if (blockIdx.x == 0 && lane == 0) g_bf3_stamps[wave][5] = work;
#endif

    // ---- epilogue (wide_conv_kernel's, on the 256 consumer threads): partial tiles -> LDS, GroupNorm + Mish + add, store
    float* Y = reinterpret_cast<float*>(lds_raw);
    constexpr int PPR = 256 / MS, ROW_F4 = LOUT * CG / 4, NF4 = (ROW_F4 + PPR - 1) / PPR;
    const int erow = (tid & 255) / PPR, epart = (tid & 255) % PPR;
    const int eb = min(b0 + erow, a_B - 1);
    float4 g4[NF4], be4[NF4], ad4[NF4];
    if (consumer) {
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int col = 4 * (epart + PPR * i);
            const int l = col / CG, ch = co0 + col % CG;
            g4[i] = *reinterpret_cast<const float4*>(p.gamma + ch);
            be4[i] = *reinterpret_cast<const float4*>(p.beta + ch);
            ad4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.add_res) ad4[i] = *reinterpret_cast<const float4*>(p.add_res + ((size_t)eb * LOUT + l) * a_Cout + ch);
            else if (p.add_tb) ad4[i] = *reinterpret_cast<const float4*>(p.add_tb + ch);
        }
        const float bias_v = (ks == 0) ? p.bias[co0 + s * 32 + (lane & 31)] : 0.0f;
        float* Yw = Y + ks * (MS * YS) + s * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            Yw[row * YS + 0 * CG] = bias_v + acc[0][r] + acc[1][r];  // y0 = P + Q
            Yw[row * YS + 1 * CG] = bias_v + acc[0][r] + acc[2][r];  // y1 = P + R
        }
    }
    __syncthreads();
    if (consumer) {
        const int b = b0 + erow;
        float4 v[NF4];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int f = epart + PPR * i;
            v[i] = *reinterpret_cast<const float4*>(Y + erow * YS + 4 * f);
#pragma unroll
            for (int q = 1; q < NP; ++q) {
                const float4 pv = *reinterpret_cast<const float4*>(Y + q * (MS * YS) + erow * YS + 4 * f);
                v[i].x += pv.x, v[i].y += pv.y, v[i].z += pv.z, v[i].w += pv.w;
            }
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        auto row_group_sum = [&](float x) __attribute__((always_inline)) {  // GS == CG: all PPR = 8 threads of the row
            x = dpp_xor_add<1>(x);
            x = dpp_xor_add<2>(x);
            x = dpp_xor_add<4>(x);
            return x;
        };
        constexpr float inv_n = 1.0f / (float)(LOUT * GS);
        const float mean = row_group_sum(sum) * inv_n;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
            sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        const float rstd = 1.0f / sqrtf(row_group_sum(sq) * inv_n + 1e-5f);
        if (b < a_B) {
#pragma unroll
            for (int i = 0; i < NF4; ++i) {
                const int col = 4 * (epart + PPR * i);
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
    BF3_STAMP(3)
}

// host: [tap][Cout][Cin] (taps 1..3 used) -> slots w2, w3 - w2, w1 - w2 (as pack_fragments_k2) -> bf16-triple fragment stream
// [Cout/32][Cin/16][slot 3][component 3][64 lanes][8 bf16]: lane (n = lane % 32, oct = lane / 32) holds
// W_comp[slot][slab * 32 + n][16 kg + 8 oct + 0..7] - the B operand of v_mfma_f32_32x32x16_bf16
inline void pack_fragments_k2_bf3(const float* w_tco_ci, int cout, int cin, unsigned short* out) {
    const size_t n = (size_t)cout * cin;
    const int nkg = cin / 16;
    for (int sl = 0; sl < cout / 32; ++sl)
        for (int kg = 0; kg < nkg; ++kg)
            for (int slot = 0; slot < 3; ++slot)
                for (int lane = 0; lane < 64; ++lane) {
                    const int nn = lane % 32, oct = lane / 32;
                    for (int j = 0; j < 8; ++j) {
                        const size_t i = (size_t)(sl * 32 + nn) * cin + 16 * kg + 8 * oct + j;
                        const float w1 = w_tco_ci[1 * n + i], w2 = w_tco_ci[2 * n + i], w3 = w_tco_ci[3 * n + i];
                        const float wv = slot == 0 ? w2 : slot == 1 ? w3 - w2 : w1 - w2;
                        unsigned c3[3];
                        split3_bf16(wv, c3[0], c3[1], c3[2]);
                        for (int comp = 0; comp < 3; ++comp)
                            out[((((size_t)sl * nkg + kg) * 3 + slot) * 3 + comp) * 512 + lane * 8 + j] = (unsigned short)c3[comp];
                    }
                }
}

inline int launch_k2_bf3(const RcbP& p, const void* w_bf3, hipStream_t s) {
    static bool attr_set = false;
    constexpr size_t bytes = Bf3Cfg::lds_bytes();
    static_assert(bytes <= 160 * 1024, "exceeds the 160 KiB LDS of a CU");
    if (!attr_set) {
        EDMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k2_bf3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        attr_set = true;
    }
    EDMP_REQUIRE(p.C1 % Bf3Cfg::KC == 0 && p.C2 % Bf3Cfg::KC == 0 && p.Cout % Bf3Cfg::CG == 0, "k2_bf3_kernel: channels must be multiples of 64");
    const int ng = p.Cout / Bf3Cfg::CG, nt = (p.B + 31) / 32;
    const int gx = xcd_split(ng, nt, (double)p.Cout * (p.C1 + p.C2) * 3 * 1.5, (double)nt * 32 * 2 * (p.C1 + p.C2));
    int gxs = -1, ngs = -1;
    if (gx > 0 && (ng & (ng - 1)) == 0) {
        gxs = __builtin_ctz(gx);
        ngs = __builtin_ctz(ng);
    }
    hipLaunchKernelGGL(k2_bf3_kernel, dim3(ng * nt), dim3(512), bytes, s, p.src1, p.src2, w_bf3, p.C1, p.C2, p.Cout, p.B, gxs, ngs, p);
    return EDMP_OK;
}

}  // namespace edmp
