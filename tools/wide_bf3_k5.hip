// wide_bf3_k5.hip — EXPERIMENT (round 4): the direct-form Conv1dBlock of the 256-channel levels (wide.hip:
// wide_conv_kernel<WK_K5, 32, 32, 32, LIN, RES>, LIN = 7 and 4) on the bf16 matrix pipe with EXACT products.
//
// As tools/wide_bf3.hip (round 3, the L = 2 instance): an fp32 number is exactly hi + mid + lo with three bf16 numbers, the product
// of two bf16 numbers is exact in fp32, so a * b = the sum of NINE exact partial products; nine v_mfma_f32_32x32x16_bf16 with fp32
// accumulation do the work of eight v_mfma_f32_32x32x2_f32 in 9 x 32 instead of 8 x 64 matrix-pipe cycles.  All nine are issued.
// Round 3 found the L = 2 instance bound by its weight stream (each weight element meets 32 samples x 2 positions).  Here every
// weight element meets 32 samples x up to LIN positions: the stream is 63 MB per launch (3.1 TB/s), the pipe is the bound.
//
// Decomposition (differs from the fp32 kernel's 4-way K split: a K slice of v_mfma_f32_32x32x16_bf16 is 16 channels and a staged
// chunk of 4 x 16 channels in three bf16 planes would not fit the LDS twice):
//   workgroup = 32 samples x 32 output channels (one GroupNorm group) x all LIN positions, 8 waves:
//   * four MFMA waves (kq, h): K16 group kq of the 32-channel chunk x tile set h - the output positions are dealt to two sets with
//     equal numbers of (output position, input position) pairs ({0,1,5,6} | {2,3,4} at LIN = 7: 14 | 15 pairs), so a weight fragment
//     is loaded by two waves and an activation fragment read by two; per pair 9 MFMAs, consecutive MFMAs on different accumulators;
//   * four staging waves: global fp32 -> split into three bf16 planes (v_perm_b32 packing) -> LDS, one chunk ahead, so the split's
//     VALU work runs beside the matrix pipe on the same SIMDs (two waves per SIMD) instead of between its instructions;
//   * LDS: [position][component][32 rows][32 + 8] bf16 per stage, two stages (107 KB);
//   * epilogue on all 512 threads: the two K-slice partial tiles per output tile -> LDS, GroupNorm(8) statistics per sample row by
//     DPP over the row's 16 threads, Mish, + time bias | residual, float4 stores.
// Weights: fragment stream of bf16 triples [Cout/32][Cin/16][slot][component][64 lanes][8 bf16], 1 KiB blocks, split once at load.
#pragma once
#include "wide_bf3.hip"

namespace edmp {

template <int LIN, bool RES>
struct K5Bf3Cfg {
    static constexpr int MS = 32, CG = 32, GS = 32, LOUT = LIN;
    static constexpr int KC = 32, KG = 16, NKQ = KC / KG;
    static constexpr int RS = KC + 8;              // bf16 per staged row (80 B: the 16 rows of a ds_read_b128 phase hit all 64 banks once)
    static constexpr int PLANE = MS * RS;
    static constexpr int STAGE = LIN * 3 * PLANE;  // bf16 per stage
    static constexpr int NTAP = 5, NSLOT = NTAP + (RES ? 1 : 0);
    static constexpr int YS = LOUT * CG + 4, NP = NKQ;
    static constexpr int NTILE = LIN + (RES ? LIN : 0);  // output tiles: conv positions, then the residual tiles (one per input position)
    // weight slot of the pair (tile, input position); -1: no such tap.  Tiles >= LIN are the folded residual 1x1 conv.
    static constexpr int slot(int tile, int lp) {
        if (tile >= LIN) return tile - LIN == lp ? NTAP : -1;
        const int t = lp - tile + 2;
        return (t >= 0 && t < NTAP) ? t : -1;
    }
    // which tile set (MFMA wave half h) owns a tile: equal pair counts
    static constexpr int owner(int tile) {
        if (tile >= LIN) return (tile - LIN) & 1;  // residual tiles alternate
        if (LIN == 7) return (tile <= 1 || tile >= 5) ? 0 : 1;
        return tile < LIN / 2 ? 0 : 1;
    }
    static constexpr int local(int tile) {  // index of the tile among its owner's tiles
        int n = 0;
        for (int q = 0; q < tile; ++q) n += owner(q) == owner(tile) ? 1 : 0;
        return n;
    }
    static constexpr int ntiles(int h) {
        int n = 0;
        for (int q = 0; q < NTILE; ++q) n += owner(q) == h ? 1 : 0;
        return n;
    }
    static constexpr int MAXT = ntiles(0) > ntiles(1) ? ntiles(0) : ntiles(1);
    static constexpr size_t lds_bytes() {
        const size_t a = 2 * (size_t)STAGE * 2, y = (size_t)NP * MS * YS * 4;
        return a > y ? a : y;
    }
};

template <int LIN, bool RES>
__global__ __launch_bounds__(512) void k5_bf3_kernel(const float* a_src1, const float* a_src2, const void* a_W, int a_C1, int a_C2, int a_Cout, int a_B,
                                                     int a_gx_shift, int a_ng_shift, RcbP p) {
    using Cf = K5Bf3Cfg<LIN, RES>;
    constexpr int MS = Cf::MS, CG = Cf::CG, GS = Cf::GS, LOUT = Cf::LOUT, KC = Cf::KC, KG = Cf::KG, RS = Cf::RS, PLANE = Cf::PLANE, STAGE = Cf::STAGE;
    constexpr int YS = Cf::YS, NP = Cf::NP, NSLOT = Cf::NSLOT, NTILE = Cf::NTILE, MAXT = Cf::MAXT;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned short* stg = reinterpret_cast<unsigned short*>(lds_raw);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    const int kq = wave & 1, h = (wave >> 1) & 1;
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

    // ---- staging side: item = (sample row, channel quad) of the chunk, all LIN positions: 32 rows x 8 quads = 256 items
    const int ptid = tid - 256;
    const int prow = (ptid >> 3) & 31, pq = ptid & 7;
    const int psb = min(b0 + prow, a_B - 1);
    auto fetch = [&](int nc, float4 (&x)[LIN]) __attribute__((always_inline)) {
        const bool first = nc < ch1;
        const float* src = first ? a_src1 : a_src2;
        const int Cs = first ? a_C1 : a_C2, c0 = (first ? nc : nc - ch1) * KC;
        const float* base = src + ((size_t)psb * LIN) * Cs + c0 + 4 * pq;
#pragma unroll
        for (int lp = 0; lp < LIN; ++lp) x[lp] = *reinterpret_cast<const float4*>(base + (size_t)lp * Cs);
    };
    auto commit = [&](unsigned short* st, const float4 (&x)[LIN]) __attribute__((always_inline)) {
#pragma unroll
        for (int lp = 0; lp < LIN; ++lp) {
            // component m = upper 16 bits of x, of x - hi, of x - hi - mid (each difference exact); v_perm_b32 packs two upper halves
            unsigned short* base = st + (lp * 3) * PLANE + prow * RS + 4 * pq;
            float x4[4] = {x[lp].x, x[lp].y, x[lp].z, x[lp].w};
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                unsigned u[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) u[j] = __builtin_bit_cast(unsigned, x4[j]);
                *reinterpret_cast<u32x2_t*>(base + m * PLANE) = u32x2_t{__builtin_amdgcn_perm(u[1], u[0], 0x07060302u), __builtin_amdgcn_perm(u[3], u[2], 0x07060302u)};
                if (m < 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) x4[j] = x4[j] - __builtin_bit_cast(float, u[j] & 0xffff0000u);
                }
            }
        }
    };

    // ---- MFMA side: weight stream of (slab = channel group, K16 group): blocks of 1 KiB [slot][component][64 lanes][16 B]
    // A chunk is worked in three PHASES, one per weight component (lo, mid, hi): phase m needs only component m of the five (six)
    // weight slots - 20 (24) registers -, and the set is refilled with the NEXT chunk's component m as soon as the phase is done:
    // every weight fragment is requested two phases (~2.9 k cycles) before its first use, with 60 registers of weights in flight
    // instead of the 120 of a whole-chunk double buffer (the budget is 256 registers at two waves per SIMD).
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(a_W) + ((size_t)grp * NKG) * (NSLOT * 3 * 1024) + 16 * lane;
    u32x4_t bw[3][NSLOT];
    auto load_w = [&](int kg, auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        const unsigned char* w = wb + (size_t)kg * (NSLOT * 3 * 1024) + m * 1024;
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) bw[m][j] = *reinterpret_cast<const u32x4_t*>(w + j * 3 * 1024);
    };
    f32x16_t acc[MAXT];
#pragma unroll
    for (int a = 0; a < MAXT; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;

    // ---- prologue
    BF3_STAMP(0)
    if (!consumer) {
        float4 xr[LIN];
        fetch(0, xr);
        commit(stg, xr);
    } else {
        load_w(kq, std::integral_constant<int, 2>{});
        load_w(kq, std::integral_constant<int, 1>{});
        load_w(kq, std::integral_constant<int, 0>{});
    }
    __syncthreads();
    BF3_STAMP(1)

    // A fragment of a lane: row lane % 32, channel octet lane / 32 of the wave's K16 group
    const int afrag = (lane & 31) * RS + 8 * (lane >> 5) + KG * kq;
    // one chunk of one tile set: phases over the weight component (lo first: small terms first), inside a phase every input position's
    // three activation components against every tile the position feeds; consecutive MFMAs on different accumulators where a
    // position feeds more than one tile; the next position's fragments are read a position ahead
    auto chunk = [&](auto hc, const unsigned short* st, int kgn) __attribute__((always_inline)) {
        constexpr int H = decltype(hc)::value;
        // fragment sets alternate between two register sets along the sequence (phase, position); the sequence of a chunk has
        // 3 * LIN entries, entry q reads entry q + 1 (the first position of the next phase included) before its MFMAs
        u32x4_t av[2][3];
#pragma unroll
        for (int m = 0; m < 3; ++m) av[0][m] = *reinterpret_cast<const u32x4_t*>(st + m * PLANE + afrag);
        static_for<0, 3>([&](auto pc) __attribute__((always_inline)) {
            constexpr int ph = decltype(pc)::value;
            constexpr int ib = 2 - ph;
            static_for<0, LIN>([&](auto lpc) __attribute__((always_inline)) {
                constexpr int lp = decltype(lpc)::value;
                constexpr int q = ph * LIN + lp;
                constexpr int lpn = (lp + 1) % LIN;
                if constexpr (q + 1 < 3 * LIN) {
#pragma unroll
                    for (int m = 0; m < 3; ++m) av[(q + 1) & 1][m] = *reinterpret_cast<const u32x4_t*>(st + (lpn * 3 + m) * PLANE + afrag);
                }
                static_for<0, 3>([&](auto ac) __attribute__((always_inline)) {
                    constexpr int ia = 2 - decltype(ac)::value;
                    static_for<0, NTILE>([&](auto lc) __attribute__((always_inline)) {
                        constexpr int tl = decltype(lc)::value;
                        if constexpr (Cf::owner(tl) == H && Cf::slot(tl, lp) >= 0) {
                            constexpr int sl = Cf::slot(tl, lp) >= 0 ? Cf::slot(tl, lp) : 0;
                            constexpr int la = Cf::local(tl);
                            acc[la] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av[q & 1][ia]), __builtin_bit_cast(bf16x8_t, bw[ib][sl]), acc[la], 0, 0, 0);
                        }
                    });
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            load_w(kgn, std::integral_constant<int, ib>{});  // this component of the next chunk (the last chunk re-reads its own: harmless)
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // one loop per role (the roles are wave-uniform; every wave passes the same nK barriers): a single-path loop body lets the
    // register allocator keep each weight set in place across iterations instead of copying "next" sets at the back edge
    auto mfma_loop = [&](auto hc) __attribute__((always_inline)) {
        for (int c = 0; c < nK; ++c) {
            chunk(hc, stg + (c & 1) * STAGE, min((c + 1) * Cf::NKQ + kq, NKG - 1));
            __syncthreads();
        }
    };
    if (consumer) {
        if (h == 0) mfma_loop(std::integral_constant<int, 0>{});
        else mfma_loop(std::integral_constant<int, 1>{});
    } else {
        for (int c = 0; c < nK; ++c) {
            if (c + 1 < nK) {
                // fetched and committed inside the step (a chunk of MFMAs lasts ~4 k cycles, the fetch ~2 k)
                float4 xr[LIN];
                fetch(c + 1, xr);
                commit(stg + ((c + 1) & 1) * STAGE, xr);
            }
            __syncthreads();
        }
    }
    BF3_STAMP(2)

    // ---- epilogue: K-slice partial tiles (+ bias in slice 0) -> LDS [kq][row][tile * CG + col]; then all 512 threads: 16 per
    //      sample row sum the partials, reduce the GroupNorm statistics over the row, normalise, Mish, add, store float4
    float* Y = reinterpret_cast<float*>(lds_raw);
    constexpr int PPR = 512 / MS, ROW_F4 = LOUT * CG / 4, NF4 = (ROW_F4 + PPR - 1) / PPR;
    const int erow = tid / PPR, epart = tid % PPR;
    const int eb = min(b0 + erow, a_B - 1);
    float4 g4[NF4], be4[NF4], ad4[NF4];
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
        const int f = min(epart + PPR * i, ROW_F4 - 1);
        const int col = 4 * f;
        const int l = col / CG, ch = co0 + col % CG;
        g4[i] = *reinterpret_cast<const float4*>(p.gamma + ch);
        be4[i] = *reinterpret_cast<const float4*>(p.beta + ch);
        ad4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.add_res) ad4[i] = *reinterpret_cast<const float4*>(p.add_res + ((size_t)eb * LOUT + l) * a_Cout + ch);
        else if (p.add_tb) ad4[i] = *reinterpret_cast<const float4*>(p.add_tb + ch);
    }
    auto spill = [&](auto hc, bool res_pass) __attribute__((always_inline)) {
        constexpr int H = decltype(hc)::value;
        const float bias_v = (kq == 0) ? (res_pass ? (RES ? p.res_bias[co0 + (lane & 31)] : 0.f) : p.bias[co0 + (lane & 31)]) : 0.0f;
        float* Yw = Y + kq * (MS * YS) + (lane & 31);
        static_for<0, NTILE>([&](auto lc) __attribute__((always_inline)) {
            constexpr int tl = decltype(lc)::value;
            if constexpr (Cf::owner(tl) == H) {
                constexpr int la = Cf::local(tl);
                if ((tl >= LIN) == res_pass) {
                    constexpr int l = tl >= LIN ? tl - LIN : tl;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        Yw[row * YS + l * CG] = acc[la][r] + bias_v;
                    }
                }
            }
        });
    };
    if constexpr (RES) {  // the folded residual 1x1 conv first: conv + res_bias -> res_out (conv2's residual addend)
        if (consumer) {
            if (h == 0) spill(std::integral_constant<int, 0>{}, true);
            else spill(std::integral_constant<int, 1>{}, true);
        }
        __syncthreads();
        if (b0 + erow < a_B) {
#pragma unroll
            for (int i = 0; i < NF4; ++i) {
                const int f = epart + PPR * i;
                if (f < ROW_F4) {
                    const int col = 4 * f;
                    const int l = col / CG, ch = co0 + col % CG;
                    float4 rv = *reinterpret_cast<const float4*>(Y + erow * YS + col);
                    const float4 pv = *reinterpret_cast<const float4*>(Y + (MS * YS) + erow * YS + col);
                    rv.x += pv.x, rv.y += pv.y, rv.z += pv.z, rv.w += pv.w;
                    *reinterpret_cast<float4*>(p.res_out + ((size_t)(b0 + erow) * LIN + l) * a_Cout + ch) = rv;
                }
            }
        }
        __syncthreads();
    }
    if (consumer) {
        if (h == 0) spill(std::integral_constant<int, 0>{}, false);
        else spill(std::integral_constant<int, 1>{}, false);
    }
    __syncthreads();
    {
        const int b = b0 + erow;
        float4 v[NF4];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int f = epart + PPR * i;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < ROW_F4) {
                v[i] = *reinterpret_cast<const float4*>(Y + erow * YS + 4 * f);
                const float4 pv = *reinterpret_cast<const float4*>(Y + (MS * YS) + erow * YS + 4 * f);
                v[i].x += pv.x, v[i].y += pv.y, v[i].z += pv.z, v[i].w += pv.w;
                sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
        }
        auto row_group_sum = [&](float x) __attribute__((always_inline)) {  // GS == CG: all 16 threads of the sample row
            x = dpp_xor_add<1>(x);
            x = dpp_xor_add<2>(x);
            x = dpp_xor_add<4>(x);
            x = dpp_xor_add<8>(x);
            return x;
        };
        constexpr float inv_n = 1.0f / (float)(LOUT * GS);
        const float mean = row_group_sum(sum) * inv_n;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int f = epart + PPR * i;
            if (f < ROW_F4) {
                const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
                sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        }
        const float rstd = 1.0f / sqrtf(row_group_sum(sq) * inv_n + 1e-5f);
        if (b < a_B) {
#pragma unroll
            for (int i = 0; i < NF4; ++i) {
                const int f = epart + PPR * i;
                if (f < ROW_F4) {
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
    }
    BF3_STAMP(3)
}

// host: [tap][Cout][Cin] (taps 0..4, tap index 5 = the folded residual 1x1 conv) -> bf16-triple fragment stream
// [Cout/32][Cin/16][nslot][component 3][64 lanes][8 bf16]: lane (n = lane % 32, oct = lane / 32) holds
// W_comp[slot][slab * 32 + n][16 kg + 8 oct + 0..7] - the B operand of v_mfma_f32_32x32x16_bf16
inline void pack_fragments_k5_bf3(const float* w_tco_ci, int cout, int cin, bool res, unsigned short* out) {
    const size_t n = (size_t)cout * cin;
    const int nkg = cin / 16, nslot = 5 + (res ? 1 : 0);
    for (int sl = 0; sl < cout / 32; ++sl)
        for (int kg = 0; kg < nkg; ++kg)
            for (int slot = 0; slot < nslot; ++slot)
                for (int lane = 0; lane < 64; ++lane) {
                    const int nn = lane % 32, oct = lane / 32;
                    for (int j = 0; j < 8; ++j) {
                        const size_t i = (size_t)(sl * 32 + nn) * cin + 16 * kg + 8 * oct + j;
                        const float wv = w_tco_ci[(size_t)slot * n + i];
                        unsigned c3[3];
                        split3_bf16(wv, c3[0], c3[1], c3[2]);
                        for (int comp = 0; comp < 3; ++comp)
                            out[((((size_t)sl * nkg + kg) * nslot + slot) * 3 + comp) * 512 + lane * 8 + j] = (unsigned short)c3[comp];
                    }
                }
}

template <int LIN, bool RES>
inline int launch_k5_bf3(const RcbP& p, const void* w_bf3, hipStream_t s) {
    static bool attr_set = false;
    using Cf = K5Bf3Cfg<LIN, RES>;
    constexpr size_t bytes = Cf::lds_bytes();
    static_assert(bytes <= 160 * 1024, "exceeds the 160 KiB LDS of a CU");
    if (!attr_set) {
        EDMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k5_bf3_kernel<LIN, RES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        attr_set = true;
    }
    EDMP_REQUIRE(p.C1 % Cf::KC == 0 && p.C2 % Cf::KC == 0 && p.Cout % Cf::CG == 0, "k5_bf3_kernel: channels must be multiples of 32");
    const int ng = p.Cout / Cf::CG, nt = (p.B + 31) / 32;
    const int gx = xcd_split(ng, nt, (double)p.Cout * (p.C1 + p.C2) * Cf::NSLOT * 1.5, (double)nt * 32 * LIN * (p.C1 + p.C2));
    int gxs = -1, ngs = -1;
    if (gx > 0 && (ng & (ng - 1)) == 0) {
        gxs = __builtin_ctz(gx);
        ngs = __builtin_ctz(ng);
    }
    hipLaunchKernelGGL((k5_bf3_kernel<LIN, RES>), dim3(ng * nt), dim3(512), bytes, s, p.src1, p.src2, w_bf3, p.C1, p.C2, p.Cout, p.B, gxs, ngs, p);
    return EDMP_OK;
}

}  // namespace edmp
