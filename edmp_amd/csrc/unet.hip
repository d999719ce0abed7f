// unet.hip — TemporalUNet eps-prediction for gfx950 (MI355X), fp32 end to end.
//
// Replaces TemporalUNet.forward and the blocks it is built from (reference: diffusion/models/temporalunet.py:47-76,
// diffusion/models/blocks.py:13-34 Conv1dBlock, :38-92 time embedding, :137-166 ResidualConvolutionBlock,
// :202-260 Down/Middle/Up samplers).  Design:
//   * activations live in HBM as [B][L][C] fp32 (channels innermost); conv weights are repacked once at load into MFMA
//     B-fragment streams (wide.hip: pack_fragments) that the kernels read straight into registers.
//   * all matrix work is exact-fp32 MFMA.  Round-2 kernel families of the full-size network (42 launches per forward):
//       wide_conv_kernel (wide.hip)   every level with >= 128 channels: the position-tile kernel - Conv1d k5 + bias +
//                                     GroupNorm + Mish + add (+ folded residual 1x1 conv), k3s2 / ConvTranspose resamplers,
//                                     Karatsuba forms at L = 2 / L = 4
//       level_kernel (level.hip)      the 32/64-channel levels: two residual blocks + resampling conv (+ final conv) per launch
//     and, for architectures those have no instance for (the tiny test networks) or EDMP_NO_FUSED / EDMP_NO_LEVEL runs, the
//     round-1 kernels of this file: conv_mfma_kernel (generic implicit GEMM), rcb_rows_kernel, rcb_block_kernel, gn_mish_kernel.
//     Taps that fall into the zero padding are never issued.
//   * the whole time-embedding MLP chain depends on t only and is precomputed for t = 1..T at load (time_table_kernel).
//   * the layer program (which kernel, which buffers) is built once in edmp_unet_load / edmp_unet_load_packed.
#include <memory>

#include "common.h"
#define EDMP_STAMPS_DEFINE 1  // unity builds with -DEDMP_STAMPS: the stamp buffer lives here
#include "params.h"

namespace edmp {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// ---------------------------------------------------------------------------------------------------------------
// parameter inventory (state-dict order; mirrors edmp_amd/weights.py:unet_param_shapes)
// ---------------------------------------------------------------------------------------------------------------
struct RawT {
    int64_t off;  // float offset in the flat blob
    int d0, d1, d2;
};
struct RawConvBlock {
    RawT w, b, gw, gb;
};
struct RawRCB {
    RawConvBlock cb[2];
    RawT tw, tb;
    bool has_res;
    RawT rw, rb;
    int cin, cout;
};
struct RawNet {
    RawT t1w, t1b, t3w, t3b;
    std::vector<RawRCB> rcbs;  // down (2 per level), middle (2), up (2 per level)
    std::vector<RawT> down_w, down_b, up_w, up_b;
    RawConvBlock final_cb;
    RawT final_w, final_b;
    int64_t total = 0;
};

static RawT take(int64_t& off, int d0, int d1 = 1, int d2 = 1) {
    RawT t{off, d0, d1, d2};
    off += (int64_t)d0 * d1 * d2;
    return t;
}
static RawConvBlock take_cb(int64_t& off, int cin, int cout, int k) {
    RawConvBlock c;
    c.w = take(off, cout, cin, k);
    c.b = take(off, cout);
    c.gw = take(off, cout);
    c.gb = take(off, cout);
    return c;
}
static RawRCB take_rcb(int64_t& off, int cin, int cout, int time_dim) {
    RawRCB r;
    r.cin = cin;
    r.cout = cout;
    r.cb[0] = take_cb(off, cin, cout, 5);
    r.cb[1] = take_cb(off, cout, cout, 5);
    r.tw = take(off, cout, time_dim);
    r.tb = take(off, cout);
    r.has_res = cin != cout;
    if (r.has_res) {
        r.rw = take(off, cout, cin, 1);
        r.rb = take(off, cout);
    }
    return r;
}
static RawNet inventory(const edmp_unet_desc& d) {
    RawNet n;
    int64_t off = 0;
    std::vector<int> dm{d.input_dim};
    for (int i = 0; i < d.n_levels; ++i) dm.push_back(d.dims[i]);
    const int td = d.time_dim;
    n.t1w = take(off, td * 4, td);
    n.t1b = take(off, td * 4);
    n.t3w = take(off, td, td * 4);
    n.t3b = take(off, td);
    const int nd = d.n_levels;
    for (int i = 0; i < nd; ++i) {
        n.rcbs.push_back(take_rcb(off, dm[i], dm[i + 1], td));
        n.rcbs.push_back(take_rcb(off, dm[i + 1], dm[i + 1], td));
        if (i != nd - 1) {
            n.down_w.push_back(take(off, dm[i + 1], dm[i + 1], 3));
            n.down_b.push_back(take(off, dm[i + 1]));
        }
    }
    n.rcbs.push_back(take_rcb(off, dm[nd], dm[nd], td));
    n.rcbs.push_back(take_rcb(off, dm[nd], dm[nd], td));
    for (int i = nd; i > 1; --i) {
        n.rcbs.push_back(take_rcb(off, dm[i] * 2, dm[i - 1], td));
        n.rcbs.push_back(take_rcb(off, dm[i - 1], dm[i - 1], td));
        n.up_w.push_back(take(off, dm[i - 1], dm[i - 1], 4));
        n.up_b.push_back(take(off, dm[i - 1]));
    }
    n.final_cb = take_cb(off, dm[1], dm[1], 5);
    n.final_w = take(off, d.input_dim, dm[1], 1);
    n.final_b = take(off, d.input_dim);
    n.total = off;
    return n;
}

// ---------------------------------------------------------------------------------------------------------------
// device program
// ---------------------------------------------------------------------------------------------------------------
// OP_RCB: Conv1dBlock of a wide level on wide_conv_kernel; OP_WRS: down/up-sampling conv of a wide level on wide_conv_kernel;
// OP_LVL: a whole 32/64-channel level (level.hip); OP_CONV / OP_GN: the generic fallback (any architecture)
enum OpKind { OP_CONV = 0, OP_GN = 1, OP_RCB = 2, OP_WRS = 4, OP_LVL = 5 };
struct Op {
    OpKind kind;
    ConvP cv;
    GnP gn;
    RcbP rc;
    LevelP lv;    // OP_LVL: a whole level
    int lv_variant;
    int lv_sb;    // samples per workgroup (level_sb, frozen at build time)
    int lv_tb1, lv_tb2;  // offsets of the two blocks' time biases in the time-bias row
    int lv_merge;        // OP_LVL: 1 = this level and the NEXT op (the following down level) run as one launch (level.hip: level2_kernel)
    int rc_L;     // OP_RCB / OP_WRS: input positions
    int rc_form;  // OP_RCB: 0 direct | 2 / 4 Karatsuba form at L = 2 / 4 (decided when the model was built)
    int rc_ms;    // OP_RCB / OP_WRS: samples per workgroup (wide_ms, frozen at build time)
    int rc_bf3;   // OP_RCB / OP_WRS: 1 = runs on the bf16 matrix pipe with exact products (bf3.hip; bf3_select, frozen at build time)
    int wrs_kind; // OP_WRS: WK_DOWN / WK_UP
    int tb_off;   // GN: offset into the time-bias row, -1 if none

    double flops_nominal, flops_exec;  // per trajectory: every tap | MFMA work actually issued (padding taps skipped, Karatsuba forms)
    double flops_direct;               // per trajectory: the direct form with padding taps skipped (round-1 'executed' accounting)
    double flops_bf16;                 // per trajectory: MFMA work issued on the bf16 pipe (bf3.hip ops: 6 partial products x the direct form; else 0)
    char name[64];                     // kernel instance as rocprofv3 prints it (without the edmp:: prefix)
};

struct UNet {
    edmp_unet_desc desc{};
    int max_batch = 0;
    float* wpack = nullptr;    // repacked conv weights + biases + gn affine
    size_t wpack_floats = 0;
    float* tbias = nullptr;    // [T][tb_stride]
    int tb_stride = 0;
    std::vector<float*> bufs;  // activation buffers
    size_t buf_cap = 0;        // floats per buffer
    std::vector<Op> prog;
    float* x_in = nullptr;   // [B][N][8]
    float* h_last = nullptr;  // [B][N][C1] input of the 1x1 head
    const float* head_w = nullptr;
    const float* head_b = nullptr;
    int head_cin = 0;
    // taps of intermediate activations for parity (buffer, C, L); valid right after a forward
    struct Tap { int which; const float* p; int C, L; };
    std::vector<Tap> taps;
    double flops_nominal = 0, flops_exec = 0, flops_direct = 0;
    double flops_bf16 = 0, flops_f32_moved = 0;  // issued on the bf16 pipe | the fp32 work (direct form) those ops replace
    int layout = 0;  // layout id of the packed weight image (Packer::layout_id)
};

// ---------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------

// (B, C, N) f32 -> [B][N][CP] f32 with zero padding of channels C..CP-1.
__global__ void pack_input_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int C, int N, int CP) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;  // over B*N*CP
    int total = B * N * CP;
    if (i >= total) return;
    int c = i % CP;
    int l = (i / CP) % N;
    int b = i / (CP * N);
    out[i] = (c < C) ? x[((size_t)b * C + c) * N + l] : 0.0f;
}


// Implicit-GEMM Conv1d / ConvTranspose1d on the fp32 MFMA.  Block = 256 threads = 4 waves, tile BM samples x BN
// output channels at ONE output position; K runs over (valid tap, source, channel chunk of KC).
// LDS tiles are [rows][KC + 4] so that the ds_read_b128 of a 16-lane group hits 16 distinct 4-bank slots.
// The valid taps of one output position form an arithmetic sequence (tap = k0 + i*dk, input position = l0 + i*dl), so
// the whole K walk is scalar arithmetic; global loads are unconditional (row indices clamped, results masked at the
// store) so that the loop body is straight-line: 4 global_load_dwordx4, 8 ds_read_b128, 16 MFMA, 4 ds_write_b128.
template <int BM, int BN, int KC>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvP p) {
    constexpr int LDK = KC + 4;
    constexpr int WN = BN / 32;
    constexpr int F4R = KC / 4;
    constexpr int A_F4 = BM * F4R;
    constexpr int B_F4 = BN * F4R;
    constexpr int A_IT = (A_F4 + 255) / 256;
    constexpr int B_IT = (B_F4 + 255) / 256;
    constexpr bool A_FULL = (A_F4 % 256) == 0;
    constexpr bool B_FULL = (B_F4 % 256) == 0;
    constexpr int STAGE = (BM + BN) * LDK;  // floats per pipeline stage: A tile then B tile
    constexpr int TS = BN + 4;              // row stride of the output tile staged for the float4 store pass
    constexpr int LDS_FL = (2 * STAGE > BM * TS) ? 2 * STAGE : BM * TS;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FL];

#define EDMP_CSTAMP(i) if constexpr (BM == 64 && BN == 64 && KC == 64) { EDMP_STAMP(6, i) }
    EDMP_CSTAMP(0)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntn = (p.Cout + BN - 1) / BN;
    const int lo = blockIdx.y / ntn;
    const int n0 = (blockIdx.y % ntn) * BN;
    const int b0 = blockIdx.x * BM;
    const int Cin = p.C1 + p.C2;

    int k0, dk, l0, dl, nvt;
    if (!p.transposed) {  // li = lo*stride + k - pad must lie in [0, Lin)
        const int base = lo * p.stride - p.pad;
        const int kmin = max(0, -base);
        const int kmax = min(p.ntaps - 1, p.Lin - 1 - base);
        k0 = kmin;
        dk = 1;
        l0 = base + kmin;
        dl = 1;
        nvt = max(0, kmax - kmin + 1);
    } else {  // li*stride + k - pad = lo  ->  k = r + i*stride, li = (lo + pad - r)/stride - i
        const int r = (lo + p.pad) % p.stride;
        const int li_first = (lo + p.pad - r) / p.stride;
        const int i_lo = max(0, li_first - (p.Lin - 1));
        const int i_hi = (p.ntaps - 1 - r) >= 0 ? min(li_first, (p.ntaps - 1 - r) / p.stride) : -1;
        k0 = r + i_lo * p.stride;
        dk = p.stride;
        l0 = li_first - i_lo;
        dl = -1;
        nvt = max(0, i_hi - i_lo + 1);
    }
    const int ch1 = p.C1 / KC, ch2 = p.C2 / KC;
    const int cpt = ch1 + ch2;  // chunks per tap
    const int nK = nvt * cpt;

    // per-thread, chunk-invariant parts of the global addresses (element offsets).  Scalars, not arrays: hipcc keeps
    // small runtime-looking arrays in scratch memory once scheduling barriers are present.
#define EDMP_DECL_A(i)                                                     \
    const int fa##i = tid + i * 256;                                       \
    const int ba##i = min(b0 + fa##i / F4R, p.B - 1);                      \
    const int a1_##i = ba##i * p.Lin * p.C1 + (fa##i % F4R) * 4;           \
    const int a2_##i = ba##i * p.Lin * p.C2 + (fa##i % F4R) * 4;           \
    const int al_##i = (fa##i / F4R) * LDK + (fa##i % F4R) * 4;            \
    const bool ap_##i = (i < A_IT) && (A_FULL || fa##i < A_F4);            \
    float4 ra##i = make_float4(0.f, 0.f, 0.f, 0.f);
#define EDMP_DECL_B(i)                                                     \
    const int fb##i = tid + i * 256;                                       \
    const int w_##i = min(n0 + fb##i / F4R, p.Cout - 1) * Cin + (fb##i % F4R) * 4; \
    const int bl_##i = BM * LDK + (fb##i / F4R) * LDK + (fb##i % F4R) * 4; \
    const bool bp_##i = (i < B_IT) && (B_FULL || fb##i < B_F4);            \
    float4 rb##i = make_float4(0.f, 0.f, 0.f, 0.f);
    EDMP_DECL_A(0) EDMP_DECL_A(1) EDMP_DECL_A(2) EDMP_DECL_A(3)
    EDMP_DECL_B(0) EDMP_DECL_B(1) EDMP_DECL_B(2) EDMP_DECL_B(3)
    static_assert(A_IT <= 4 && B_IT <= 4, "tile too large for the staging code");
    int ld_ti = 0, ld_cc = 0;  // next chunk to fetch (block-uniform)

#define EDMP_LDA(i) \
    if (ap_##i) ra##i = *reinterpret_cast<const float4*>(abase_ + (first_ ? a1_##i : a2_##i));
#define EDMP_LDB(i) \
    if (bp_##i) rb##i = *reinterpret_cast<const float4*>(wbase_ + w_##i);
// issue the global loads of chunk (ld_ti, ld_cc) and advance the chunk cursor
#define EDMP_LOAD_NEXT()                                                                                      \
    {                                                                                                         \
        const int tap_ = k0 + ld_ti * dk;                                                                     \
        const int li_ = l0 + ld_ti * dl;                                                                      \
        const bool first_ = ld_cc < ch1;                                                                      \
        const float* src_ = first_ ? p.src1 : p.src2;                                                         \
        const int Cs_ = first_ ? p.C1 : p.C2;                                                                 \
        const int ci0_ = (first_ ? ld_cc : ld_cc - ch1) * KC;                                                 \
        const float* abase_ = src_ + (size_t)li_ * Cs_ + ci0_;                                                \
        const float* wbase_ = p.W + (size_t)tap_ * p.Cout * Cin + (first_ ? 0 : p.C1) + ci0_;                 \
        EDMP_LDA(0) EDMP_LDA(1) EDMP_LDA(2) EDMP_LDA(3) EDMP_LDB(0) EDMP_LDB(1) EDMP_LDB(2) EDMP_LDB(3)                              \
        if (++ld_cc == cpt) {                                                                                 \
            ld_cc = 0;                                                                                        \
            ++ld_ti;                                                                                          \
        }                                                                                                     \
    }
#define EDMP_STA(i) \
    if (ap_##i) *reinterpret_cast<float4*>(st_ + al_##i) = ra##i;
#define EDMP_STB(i) \
    if (bp_##i) *reinterpret_cast<float4*>(st_ + bl_##i) = rb##i;
#define EDMP_STORE_STAGE(buf)                                                     \
    {                                                                             \
        float* st_ = lds + (buf) * STAGE;                                         \
        EDMP_STA(0) EDMP_STA(1) EDMP_STA(2) EDMP_STA(3) EDMP_STB(0) EDMP_STB(1) EDMP_STB(2) EDMP_STB(3) \
    }

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    // requested now, used by the store epilogue: loaded there it would expose a full memory round trip
    const float bias_v = p.bias[min(n0 + wn * 32 + (lane & 31), p.Cout - 1)];

    const int arow = (wm * 32 + (lane & 31)) * LDK + 4 * (lane >> 5);
    const int brow = BM * LDK + (wn * 32 + (lane & 31)) * LDK + 4 * (lane >> 5);
    if (nK > 0) {
        EDMP_LOAD_NEXT();
        EDMP_STORE_STAGE(0);
        __syncthreads();
        EDMP_CSTAMP(1)
        // steady state: every iteration prefetches chunk kk+1 while the MFMAs consume chunk kk
        for (int kk = 0; kk < nK - 1; ++kk) {
            const int cur = kk & 1;
            EDMP_LOAD_NEXT();
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs (hipcc otherwise sinks it)
            const float* a_s = lds + cur * STAGE + arow;
            const float* b_s = lds + cur * STAGE + brow;
#pragma unroll
            for (int j = 0; j < KC / 8; ++j) {
                const float4 a4 = *reinterpret_cast<const float4*>(a_s + 8 * j);
                const float4 b4 = *reinterpret_cast<const float4*>(b_s + 8 * j);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            EDMP_STORE_STAGE(cur ^ 1);
            __syncthreads();
        }
        {  // last chunk: nothing left to prefetch
            const int cur = (nK - 1) & 1;
            const float* a_s = lds + cur * STAGE + arow;
            const float* b_s = lds + cur * STAGE + brow;
#pragma unroll
            for (int j = 0; j < KC / 8; ++j) {
                const float4 a4 = *reinterpret_cast<const float4*>(a_s + 8 * j);
                const float4 b4 = *reinterpret_cast<const float4*>(b_s + 8 * j);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
            }
        }
    }
#undef EDMP_LOAD_NEXT
#undef EDMP_STORE_STAGE
#undef EDMP_LDA
#undef EDMP_LDB
#undef EDMP_STA
#undef EDMP_STB
#undef EDMP_DECL_A
#undef EDMP_DECL_B

    EDMP_CSTAMP(2)
    // epilogue: + bias, store [b][lo][co].  acc[r]: row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31.
    // Sixteen dword stores per lane are store-issue-bound (~4.8 k cycles measured); the tile is transposed through LDS
    // instead and leaves as BM*BN/1024 float4 stores per thread (128-byte runs along the channels).
    const int co = n0 + wn * 32 + (lane & 31);
    if ((p.Cout & 3) == 0) {
        __syncthreads();  // every wave is done with the stages
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            lds[(wm * 32 + row) * TS + wn * 32 + (lane & 31)] = acc[r] + bias_v;
        }
        __syncthreads();
        constexpr int NF = (BM * (BN / 4) + 255) / 256;
#pragma unroll
        for (int it = 0; it < NF; ++it) {
            const int f = tid + it * 256;
            const int row = f / (BN / 4), c4 = (f % (BN / 4)) * 4;
            const int b = b0 + row;
            if (f < BM * (BN / 4) && b < p.B && n0 + c4 < p.Cout)
                *reinterpret_cast<float4*>(p.dst + ((size_t)b * p.Lout + lo) * p.Cout + n0 + c4) = *reinterpret_cast<const float4*>(lds + row * TS + c4);
        }
    } else if (co < p.Cout) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            int b = b0 + wm * 32 + row;
            if (b < p.B) p.dst[((size_t)b * p.Lout + lo) * p.Cout + co] = acc[r] + bias_v;
        }
    }
    EDMP_CSTAMP(3)
#undef EDMP_CSTAMP
}

}  // namespace edmp
#include "wide.hip"
#include "bf3.hip"
#include "level.hip"
#ifdef EDMP_SHARDED  // the position-tile and whole-level kernels are compiled in parallel translation units (kernel_shard.hip)
#include "kernel_instances.h"
namespace edmp {
#define EDMP_X(sh, K, MS, CG, GS, L, R) extern template int launch_wide_t<K, MS, CG, GS, L, R>(const RcbP&, hipStream_t);
EDMP_WIDE_INSTANCES(EDMP_X)
#undef EDMP_X
#define EDMP_X(sh, K, MS, CG, GS, L, R) extern template int launch_bf3_t<K, MS, CG, GS, L, R>(const RcbP&, hipStream_t);
EDMP_BF3_INSTANCES(EDMP_X)
#undef EDMP_X
#define EDMP_X(sh, M, C, L, SB, CIN) extern template int launch_level_t<M, C, L, SB, CIN>(const LevelP&, hipStream_t);
EDMP_LEVEL_INSTANCES(EDMP_X)
#undef EDMP_X
#define EDMP_X(sh, MA, CA, LA, CINA, MB, CB, LB, CINB, SB) extern template int launch_level2_t<MA, CA, LA, CINA, MB, CB, LB, CINB, SB>(const LevelP&, const LevelP&, hipStream_t);
EDMP_LEVEL2_INSTANCES(EDMP_X)
#undef EDMP_X
}  // namespace edmp
#endif
namespace edmp {

// GroupNorm(8 groups, eps 1e-5, biased variance) -> Mish -> (+ time bias[c] | + residual[b,l,c]) in place.
// One wave per (sample, group); the (C/8) x L elements of the group stay in registers between the passes.
template <int EPL>  // elements per lane
__global__ __launch_bounds__(256) void gn_mish_kernel(GnP p) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);  // global wave = b*8 + g
    if (gw >= p.B * 8) return;
    const int b = gw >> 3, g = gw & 7;
    const int cg = p.C >> 3;
    const int n = cg * p.L;
    float* base = p.y + (size_t)b * p.L * p.C + g * cg;
    float v[EPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        int e = lane + i * 64;
        float x = 0.f;
        if (e < n) {
            int l = e / cg, c = e - l * cg;
            x = base[(size_t)l * p.C + c];
        }
        v[i] = x;
        s += x;
    }
    const float mean = wave_sum(s) / (float)n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        int e = lane + i * 64;
        float d = (e < n) ? (v[i] - mean) : 0.f;
        q += d * d;
    }
    const float var = wave_sum(q) / (float)n;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        int e = lane + i * 64;
        if (e < n) {
            int l = e / cg, c = e - l * cg;
            int ch = g * cg + c;
            float scale = rstd * p.gamma[ch];
            float shift = p.beta[ch] - scale * mean;
            float y = mish_fast(v[i] * scale + shift);
            if (p.add_tbias) y += p.add_tbias[ch];
            if (p.add_res) y += p.add_res[((size_t)b * p.L + l) * p.C + ch];
            base[(size_t)l * p.C + c] = y;
        }
    }
}

// final 1x1 conv (final_conv.1, temporalunet.py:36): h [B][N][Cin] -> eps (B, Cout, N) in the reference layout.
__global__ void head_1x1_kernel(const float* __restrict__ h, const float* __restrict__ w, const float* __restrict__ bias,
                                float* __restrict__ eps, int B, int N, int Cin, int Cout) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;  // over B*N
    if (i >= B * N) return;
    int b = i / N, l = i - b * N;
    const float* hp = h + (size_t)i * Cin;
    for (int co = 0; co < Cout; ++co) {
        float a = bias[co];
        for (int ci = 0; ci < Cin; ++ci) a = fmaf(hp[ci], w[co * Cin + ci], a);
        eps[((size_t)b * Cout + co) * N + l] = a;
    }
}

// [B][L][C] -> (B, C, L)   (parity/debug only)
__global__ void unpack_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int L, int C) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * L * C) return;
    int l = i % L;
    int c = (i / L) % C;
    int b = i / (L * C);
    dst[i] = src[((size_t)b * L + l) * C + c];
}

// time-bias table: for t = 1..T (blockIdx.x = t-1): temb = Linear(Mish(Linear(sinusoid(t))));  row[c] = tb[c] +
// sum_k tw[c][k] * Mish(temb[k]) for the concatenated per-RCB time MLPs.  blocks.py:46-54, 83-88, 64-67.
__global__ void time_table_kernel(const float* __restrict__ t1w, const float* __restrict__ t1b, const float* __restrict__ t3w,
                                  const float* __restrict__ t3b, const float* __restrict__ tw, const float* __restrict__ tb,
                                  float* __restrict__ out, int time_dim, int stride) {
    extern __shared__ float sm[];  // emb[time_dim] | hid[4*time_dim] | temb[time_dim]
    float* emb = sm;
    float* hid = sm + time_dim;
    float* temb = hid + 4 * time_dim;
    const float t = (float)(blockIdx.x + 1);
    const int half = time_dim / 2;
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        float e = (float)(log(10000.0) / (double)(half - 1));  // python float, then cast when multiplied into f32
        float f = expf((float)i * -e);
        float a = t * f;
        emb[i] = sinf(a);
        emb[i + half] = cosf(a);
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 4 * time_dim; o += blockDim.x) {
        float a = t1b[o];
        for (int k = 0; k < time_dim; ++k) a = fmaf(emb[k], t1w[o * time_dim + k], a);
        hid[o] = mish_f(a);
    }
    __syncthreads();
    for (int o = threadIdx.x; o < time_dim; o += blockDim.x) {
        float a = t3b[o];
        for (int k = 0; k < 4 * time_dim; ++k) a = fmaf(hid[k], t3w[o * 4 * time_dim + k], a);
        temb[o] = mish_f(a);  // every consumer applies Mish first (TimeMLP, blocks.py:64-67)
    }
    __syncthreads();
    for (int c = threadIdx.x; c < stride; c += blockDim.x) {
        float a = tb[c];
        for (int k = 0; k < time_dim; ++k) a = fmaf(temb[k], tw[(size_t)c * time_dim + k], a);
        out[(size_t)blockIdx.x * stride + c] = a;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host: load / build program
// ---------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int KC>
static void launch_conv_t(const ConvP& p, hipStream_t s) {
    dim3 grid((p.B + BM - 1) / BM, p.Lout * ((p.Cout + BN - 1) / BN));
    hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, KC>), grid, dim3(256), 0, s, p);
}
static int pick_kc(const ConvP& p) {
    auto ok = [&](int kc) { return p.C1 % kc == 0 && (p.C2 == 0 || p.C2 % kc == 0); };
    return ok(64) ? 64 : ok(32) ? 32 : ok(16) ? 16 : 8;
}
static void launch_conv(const ConvP& p, hipStream_t s) {
    const int kc = pick_kc(p);
    const bool wide = (p.Cout % 64 == 0);
    if (wide) {
        if (kc == 64) launch_conv_t<64, 64, 64>(p, s);
        else if (kc == 32) launch_conv_t<64, 64, 32>(p, s);
        else if (kc == 16) launch_conv_t<64, 64, 16>(p, s);
        else launch_conv_t<64, 64, 8>(p, s);
    } else {
        if (kc >= 32) launch_conv_t<128, 32, 32>(p, s);
        else if (kc == 16) launch_conv_t<128, 32, 16>(p, s);
        else launch_conv_t<128, 32, 8>(p, s);
    }
}
static int launch_gn(const GnP& p, hipStream_t s) {
    const int n = (p.C / 8) * p.L;
    dim3 grid((p.B * 8 + 3) / 4);
    if (n <= 128) hipLaunchKernelGGL((gn_mish_kernel<2>), grid, dim3(256), 0, s, p);
    else if (n <= 256) hipLaunchKernelGGL((gn_mish_kernel<4>), grid, dim3(256), 0, s, p);
    else if (n <= 512) hipLaunchKernelGGL((gn_mish_kernel<8>), grid, dim3(256), 0, s, p);
    else {
        set_error("GroupNorm group of %d elements exceeds the register-resident limit (512)", n);
        return EDMP_ERR_ARG;
    }
    return EDMP_OK;
}

// Builder switches, read from the environment WHEN A MODEL IS BUILT (edmp_unet_load*) and frozen into its layer program:
// two models built under different settings can live side by side in one process (A/B runs, the adversarial-weights test).
// EDMP_NO_KARATSUBA=1: the L = 2 convolutions of the 512-channel levels in the direct form instead of the Karatsuba form
// (3 instead of 4 matrix products; wide.hip WK_K5K2) and the L = 4 convolutions of the 256/512-channel levels (nested form, 9 instead
// of 14 products; WK_K5K4).
static bool karatsuba_l2() { return getenv("EDMP_NO_KARATSUBA") == nullptr; }
static bool karatsuba_l4() { return karatsuba_l2(); }
static bool rcb_supported(int cout, int L, int c1, int c2) {
    const int cg = cout / 8;
    const bool shape = (cg == 64 && (L == 2 || L == 4)) || (cg == 32 && (L == 4 || L == 7)) || (cg == 16 && (L == 7 || L == 13));
    return shape && cout % 8 == 0 && c1 % 32 == 0 && c2 % 32 == 0 && (c2 == 0 || c2 == c1);  // wide.hip: equal halves of a concat
}
// bf16x3 split (bf3.hip): the direct-form instances whose weights meet >= 7 positions run on the bf16 matrix pipe with exact
// products and fp32 accumulation - measured x1.2-1.55 per launch at HALF the fp32-MFMA kernel's error against float64
// (profiles/r06_bf16x3.md).  EDMP_BF16X3=<mask> (read at model-build time, frozen into the layer program and the packed image's
// layout id): bit 0 Conv1dBlock L = 7 / 256 ch, bit 1 Conv1dBlock L = 7 / 128 ch, bit 2 Conv1dBlock L = 13 / 128 ch,
// bit 3 the k3s2 (L = 7) / ConvTranspose (L = 4) resamplers at 256 channels, bit 4 those at 512 channels (L = 4 / L = 2), bit 5 those at
// 128 channels (L = 13 / L = 7), bit 6 Conv1dBlock L = 4 / 512 ch, bit 7 Conv1dBlock L = 4 / 256 ch (direct form on the bf16 pipe instead of the
// nested Karatsuba form on the fp32 pipe).  0 = every conv on the fp32-MFMA kernels (bench.py's A/B leg).
static const int kBf3Default = 0xff;  // same-box bench A/Bs (profiles/r06_bf16x3.md): 0 -> 1 001 k, 0x7 -> 1 083 k, 0x3f -> 1 102 k traj-steps/s; another box: 0x3f 1 067 k, 0x7f 1 101 k, 0xbf 1 087 k, 0xff -> 1 120 k
static int bf3_mask() {
    const char* e = getenv("EDMP_BF16X3");
    return e ? (int)strtol(e, nullptr, 0) : kBf3Default;
}
// kind: WK_K5 (a Conv1dBlock; L = its length), WK_DOWN / WK_UP (L = input length)
static bool bf3_select(int cout, int L, int kind) {
    const int cg = cout / 8, m = bf3_mask();
    if (kind == WK_K5) {
        if (cg == 32 && L == 7) return m & 1;
        if (cg == 16 && L == 7) return m & 2;
        if (cg == 16 && L == 13) return m & 4;
        if (cg == 64 && L == 4) return m & 64;
        if (cg == 32 && L == 4) return m & 128;
    }
    if (kind == WK_DOWN) {
        if (cg == 32 && L == 7) return m & 8;
        if (cg == 64 && L == 4) return m & 16;
        if (cg == 16 && L == 13) return m & 32;
    }
    if (kind == WK_UP) {
        if (cg == 32 && L == 4) return m & 8;
        if (cg == 64 && L == 2) return m & 16;
        if (cg == 16 && L == 7) return m & 32;
    }
    return false;
}
// form: 0 direct | 2 Karatsuba at L = 2 | 4 nested Karatsuba at L = 4 (rcb_form, frozen into the op at build time)
static int rcb_form(int cout, int L) {
    const int cg = cout / 8;
    if (bf3_select(cout, L, WK_K5)) return 0;  // bf3.hip runs the direct form
    if (cg == 64 && L == 2 && karatsuba_l2()) return 2;
    if (cg >= 32 && L == 4 && karatsuba_l4()) return 4;
    return 0;
}
// position-tile kernel (wide.hip) instances: the tile height MS = samples per workgroup.  16 for the 128-channel levels (32-sample
// workgroups would leave half the CUs idle); 32 for the Karatsuba forms of the 256 / 512-channel levels (their weight stream per
// FLOP doubles with 16-sample tiles: 16 TB/s out of the L2s at L = 4); and for the DIRECT-form instances of the 256 / 512-channel
// levels - Conv1dBlock at L = 7, the k3s2 / ConvTranspose resamplers - 16 since round 5: 512 workgroups per launch, two co-resident per CU
// (160 registers, 48 KB of LDS), one workgroup's prologue / epilogue / barriers run under the other's fp32 MFMAs (a wave's own VALU work
// cannot: profiles/r05_coissue_control.md).  Same-stream A/B on isolated layer chains x1.06-1.08 (tools/dualbench.hip, profiles/r05_forkjoin.md).
// EDMP_MS16=<mask> (read at model-build time, frozen into the layer program and the packed image's layout id) selects the families:
// bit 0 Conv1dBlock L = 7 / 256 ch, bit 1 k3s2 L = 7 / 256 ch, bit 2 ConvTranspose L = 4 / 256 ch, bit 3 k3s2 L = 4 / 512 ch,
// bit 4 ConvTranspose L = 2 / 512 ch.
static const int kMs16Default = 0x05;  // same-box bench A/B, three alternated runs each: 986.5 k -> 989.8 k traj-steps/s (profiles/r05_forkjoin.md); bits 1, 3, 4 measured x0.98-1.01 per launch: off
static int ms16_mask() {
    const char* e = getenv("EDMP_MS16");
    return e ? (int)strtol(e, nullptr, 0) : kMs16Default;
}
// kind: WK_K5 (a Conv1dBlock; L = its length), WK_DOWN / WK_UP (L = input length)
static int wide_ms(int cout, int L, int kind) {
    const int cg = cout / 8;
    if (bf3_select(cout, L, kind)) return cg >= 32 ? 32 : 16;  // bf3.hip: 32-sample workgroups at 256 channels (256 workgroups), 16 at 128
    if (cg < 32) return 16;
    if (kind == WK_K5 && rcb_form(cout, L) != 0) return 32;
    const int m = ms16_mask();
    if (kind == WK_K5 && cg == 32 && L == 7) return (m & 1) ? 16 : 32;
    if (kind == WK_DOWN && cg == 32 && L == 7) return (m & 2) ? 16 : 32;
    if (kind == WK_UP && cg == 32 && L == 4) return (m & 4) ? 16 : 32;
    if (kind == WK_DOWN && cg == 64 && L == 4) return (m & 8) ? 16 : 32;
    if (kind == WK_UP && cg == 64 && L == 2) return (m & 16) ? 16 : 32;
    return 32;
}
static int launch_rcb(const RcbP& p, int L, int form, int ms, int bf3, hipStream_t s) {
    const int cg = p.Cout / 8;
    const bool res = p.res_out != nullptr;
    if (bf3) {
#define EDMP_B3(MS, GS, LL) \
    return res ? launch_bf3_t<WK_K5, MS, 32, GS, LL, true>(p, s) : launch_bf3_t<WK_K5, MS, 32, GS, LL, false>(p, s);
        if (cg == 32 && L == 7) { EDMP_B3(32, 32, 7) }
        if (cg == 16 && L == 7) { EDMP_B3(16, 16, 7) }
        if (cg == 16 && L == 13) { EDMP_B3(16, 16, 13) }
        if (cg == 32 && L == 4) { EDMP_B3(32, 32, 4) }
#undef EDMP_B3
        if (cg == 64 && L == 4) return res ? launch_bf3_t<WK_K5, 32, 64, 64, 4, true>(p, s) : launch_bf3_t<WK_K5, 32, 64, 64, 4, false>(p, s);
        set_error("no bf16x3 conv+GroupNorm kernel for Cout=%d L=%d", p.Cout, L);
        return EDMP_ERR_STATE;
    }
#define EDMP_K5(MS, CG, GS, LL) \
    return res ? launch_wide_t<WK_K5, MS, CG, GS, LL, true>(p, s) : launch_wide_t<WK_K5, MS, CG, GS, LL, false>(p, s);
    if (cg == 64 && L == 2) {
        if (form == 2) return res ? launch_wide_t<WK_K5K2, 32, 64, 64, 2, true>(p, s) : launch_wide_t<WK_K5K2, 32, 64, 64, 2, false>(p, s);
        EDMP_K5(32, 64, 64, 2)
    }
    if (cg == 64 && L == 4) {
        if (form == 4) return res ? launch_wide_t<WK_K5K4, 32, 64, 64, 4, true>(p, s) : launch_wide_t<WK_K5K4, 32, 64, 64, 4, false>(p, s);
        EDMP_K5(32, 64, 64, 4)
    }
    if (cg == 32 && L == 4) {
        if (form == 4) return res ? launch_wide_t<WK_K5K4, 32, 32, 32, 4, true>(p, s) : launch_wide_t<WK_K5K4, 32, 32, 32, 4, false>(p, s);
        EDMP_K5(32, 32, 32, 4)
    }
    if (cg == 32 && L == 7 && ms == 16) { EDMP_K5(16, 32, 32, 7) }
    if (cg == 32 && L == 7) { EDMP_K5(32, 32, 32, 7) }
    if (cg == 16 && L == 7) { EDMP_K5(16, 32, 16, 7) }
    if (cg == 16 && L == 13) { EDMP_K5(16, 32, 16, 13) }
#undef EDMP_K5
    set_error("no fused conv+GroupNorm kernel for Cout=%d L=%d", p.Cout, L);
    return EDMP_ERR_STATE;
}

// whole-level kernel variants (level.hip): (mode, channels, length, stored input channels) -> id, 0 = none
static int level_variant(int mode, int C, int L, int c1, int c2) {
    if (mode == LV_DOWN && C == 32 && L == 50 && c1 == 8 && c2 == 0) return 1;
    if (mode == LV_DOWN && C == 64 && L == 25 && c1 == 32 && c2 == 0) return 2;
    if (mode == LV_UP && C == 64 && L == 13 && c1 == 128 && c2 == 128) return 3;
    if (mode == LV_UP_FINAL && C == 32 && L == 25 && c1 == 64 && c2 == 64) return 4;
    return 0;
}
static int level_kx(int variant) { return variant == 1 ? 16 : variant == 2 ? 32 : variant == 3 ? 256 : 128; }
// samples per workgroup of the level kernels, per variant: 4 = one workgroup per CU at B = 1024; 2 = two co-resident
// workgroups per CU (two waves per SIMD: one workgroup's GroupNorm / Mish epilogues, barriers and staging run under the other's
// MFMAs).  EDMP_LEVEL_SB=<d1><d2><d3><d4> (digits 2 / 4 for variants 1..4) overrides at model-build time (A/B runs).
static int level_sb(int variant) {
    static const char kDefault[] = "4222";  // same-box A/B (scripts/ab_models.py, round 3): -1.2 / -1.6 / -2.8 us per launch for variants 2 / 3 / 4, nothing for 1
    const char* e = getenv("EDMP_LEVEL_SB");
    const char* t = (e && strlen(e) == 4) ? e : kDefault;
    return t[variant - 1] == '2' ? 2 : 4;
}
// EDMP_LEVEL_MERGE=<mask> (read at model-build time): bit 0 = the two down levels of the 32/64-channel resolutions (variants 1 + 2) as
// ONE launch, level 1's k3s2 output handed to level 2 in LDS (level.hip: level2_kernel; two samples per workgroup); bit 1 = the two
// last up levels (variants 3 + 4) likewise, the ConvTranspose output of the 64-channel level going into the first half of the last
// level's input tile (the skip half still comes from HBM)
static const int kLevelMergeDefault = 0x3;  // same-box bench A/B, three alternated runs each: bit 0 986.5 k -> 988.2 k; bits 0 + 1 993.8 k -> 997.7 k, bit 1 alone 992.5 k (profiles/r05_level_kernels.md)
static int level_merge_mask() {
    const char* e = getenv("EDMP_LEVEL_MERGE");
    return e ? (int)strtol(e, nullptr, 0) : kLevelMergeDefault;
}
static int launch_level2(const LevelP& pa, const LevelP& pb, int va, int vb, hipStream_t s) {
    if (va == 1 && vb == 2) return launch_level2_t<LV_DOWN, 32, 50, 8, LV_DOWN, 64, 25, 32, 2>(pa, pb, s);
    if (va == 3 && vb == 4) return launch_level2_t<LV_UP, 64, 13, 256, LV_UP_FINAL, 32, 25, 128, 2>(pa, pb, s);
    set_error("no merged whole-level kernel for variants %d + %d", va, vb);
    return EDMP_ERR_STATE;
}
static int launch_level(const LevelP& p, int variant, int sb, hipStream_t s) {
    switch (variant * 10 + sb) {
        case 14: return launch_level_t<LV_DOWN, 32, 50, 4, 8>(p, s);
        case 24: return launch_level_t<LV_DOWN, 64, 25, 4, 32>(p, s);
        case 34: return launch_level_t<LV_UP, 64, 13, 4, 256>(p, s);
        case 44: return launch_level_t<LV_UP_FINAL, 32, 25, 4, 128>(p, s);
        case 12: return launch_level_t<LV_DOWN, 32, 50, 2, 8>(p, s);
        case 22: return launch_level_t<LV_DOWN, 64, 25, 2, 32>(p, s);
        case 32: return launch_level_t<LV_UP, 64, 13, 2, 256>(p, s);
        case 42: return launch_level_t<LV_UP_FINAL, 32, 25, 2, 128>(p, s);
    }
    set_error("no whole-level kernel variant %d", variant);
    return EDMP_ERR_STATE;
}

// down/up-sampling convs of the wide levels (no GroupNorm behind them) on the position-tile kernel
static bool wrs_supported(int cout, int cin, int Lin, bool transposed) {
    const int cg = cout / 8;
    if (cout % 8 != 0 || cin % 32 != 0) return false;
    if (transposed) return (cg == 64 && Lin == 2) || (cg == 32 && Lin == 4) || (cg == 16 && Lin == 7);
    return (cg == 64 && Lin == 4) || (cg == 32 && Lin == 7) || (cg == 16 && Lin == 13);
}
static int launch_wrs(const RcbP& p, int kind, int Lin, int ms, int bf3, hipStream_t s) {
    const int cg = p.Cout / 8;
    if (bf3) {  // bf3.hip: no GroupNorm behind a resampler, so 32-channel workgroups at every width
        if (kind == WK_DOWN && cg == 32 && Lin == 7) return launch_bf3_t<WK_DOWN, 32, 32, 32, 7, false>(p, s);
        if (kind == WK_UP && cg == 32 && Lin == 4) return launch_bf3_t<WK_UP, 32, 32, 32, 4, false>(p, s);
        if (kind == WK_DOWN && cg == 64 && Lin == 4) return launch_bf3_t<WK_DOWN, 32, 32, 32, 4, false>(p, s);
        if (kind == WK_UP && cg == 64 && Lin == 2) return launch_bf3_t<WK_UP, 32, 32, 32, 2, false>(p, s);
        if (kind == WK_DOWN && cg == 16 && Lin == 13) return launch_bf3_t<WK_DOWN, 16, 32, 16, 13, false>(p, s);
        if (kind == WK_UP && cg == 16 && Lin == 7) return launch_bf3_t<WK_UP, 16, 32, 16, 7, false>(p, s);
        set_error("no bf16x3 resampling kernel for kind=%d Cout=%d Lin=%d", kind, p.Cout, Lin);
        return EDMP_ERR_STATE;
    }
    if (ms == 16 && cg >= 32) {  // 16-sample tiles at 256 / 512 channels (wide_ms)
        if (kind == WK_DOWN && cg == 32 && Lin == 7) return launch_wide_t<WK_DOWN, 16, 32, 32, 7, false>(p, s);
        if (kind == WK_UP && cg == 32 && Lin == 4) return launch_wide_t<WK_UP, 16, 32, 32, 4, false>(p, s);
        if (kind == WK_DOWN && cg == 64 && Lin == 4) return launch_wide_t<WK_DOWN, 16, 64, 64, 4, false>(p, s);
        if (kind == WK_UP && cg == 64 && Lin == 2) return launch_wide_t<WK_UP, 16, 64, 64, 2, false>(p, s);
    }
    if (kind == WK_DOWN) {
        if (cg == 64 && Lin == 4) return launch_wide_t<WK_DOWN, 32, 64, 64, 4, false>(p, s);
        if (cg == 32 && Lin == 7) return launch_wide_t<WK_DOWN, 32, 32, 32, 7, false>(p, s);
        if (cg == 16 && Lin == 13) return launch_wide_t<WK_DOWN, 16, 32, 16, 13, false>(p, s);
    } else {
        if (cg == 64 && Lin == 2) return launch_wide_t<WK_UP, 32, 64, 64, 2, false>(p, s);
        if (cg == 32 && Lin == 4) return launch_wide_t<WK_UP, 32, 32, 32, 4, false>(p, s);
        if (cg == 16 && Lin == 7) return launch_wide_t<WK_UP, 16, 32, 16, 7, false>(p, s);
    }
    set_error("no position-tile resampling kernel for kind=%d Cout=%d Lin=%d", kind, p.Cout, Lin);
    return EDMP_ERR_STATE;
}

// kernel instance an op launches, spelled as rocprofv3's kernel trace prints it (minus the edmp:: prefix): lets
// bench.py's per-kernel table be checked line by line against profiles/*_kernel_stats.csv
static void op_kernel_name(const Op& op, char* out) {
    if (op.kind == OP_RCB || op.kind == OP_WRS) {
        const int cg = op.rc.Cout / 8, ms = op.rc_ms;
        const int kind = op.kind == OP_RCB ? (op.rc_form == 2 ? 3 : op.rc_form == 4 ? 4 : 0) : op.wrs_kind;
        if (op.rc_bf3 && op.kind == OP_WRS) snprintf(out, 64, "bf3_conv_kernel<%d, %d, 32, %d, %d, false>", kind, ms, cg < 32 ? 16 : 32, op.rc_L);
        else
        snprintf(out, 64, "%s_conv_kernel<%d, %d, %d, %d, %d, %s>", op.rc_bf3 ? "bf3" : "wide", kind, ms, cg < 32 ? 32 : cg, cg, op.rc_L,
                 (op.kind == OP_RCB && op.rc.res_out) ? "true" : "false");
    }
    else if (op.kind == OP_LVL && op.lv_merge) snprintf(out, 64, op.lv_variant == 1 ? "level2_kernel<0, 32, 50, 8, 0, 64, 25, 32, 2>" : "level2_kernel<1, 64, 13, 256, 2, 32, 25, 128, 2>");
    else if (op.kind == OP_LVL) {
        static const char* lv_fmt[] = {"", "level_kernel<0, 32, 50, %d, 8>", "level_kernel<0, 64, 25, %d, 32>", "level_kernel<1, 64, 13, %d, 256>", "level_kernel<2, 32, 25, %d, 128>"};
        snprintf(out, 64, lv_fmt[op.lv_variant], op.lv_sb);
    }
    else if (op.kind == OP_CONV) {
        const int kc = pick_kc(op.cv);
        if (op.cv.Cout % 64 == 0) snprintf(out, 64, "conv_mfma_kernel<64, 64, %d>", kc);
        else snprintf(out, 64, "conv_mfma_kernel<128, 32, %d>", kc >= 32 ? 32 : kc);
    } else {
        const int n = (op.gn.C / 8) * op.gn.L;
        snprintf(out, 64, "gn_mish_kernel<%d>", n <= 128 ? 2 : n <= 256 ? 4 : 8);
    }
}


bool unet_complete(const UNet* u) { return u && u->wpack && u->tbias && !u->prog.empty(); }

void unet_destroy(UNet* u) {
    if (!u) return;
    if (u->wpack) (void)hipFree(u->wpack);
    if (u->tbias) (void)hipFree(u->tbias);
    for (float* b : u->bufs) (void)hipFree(b);
    delete u;
}

static inline int round8(int c) { return (c + 7) / 8 * 8; }

struct Packer {
    std::vector<float> host;
    bool dry = false;   // layout only: offsets and sizes are computed, nothing is written (loading a packed image)
    size_t total = 0;   // floats packed so far (== host.size() unless dry)
    // signature of the layout actually produced: every tensor's packing form and size, in order (FNV-1a).  The builder's
    // run-time switches (EDMP_NO_FUSED, EDMP_NO_RESFOLD, EDMP_NO_LEVEL, EDMP_NO_KARATSUBA, EDMP_BF16X3, EDMP_MS16, EDMP_LEVEL_MERGE, EDMP_LEVEL_SB) move
    // tensors or change their fragment order, several of them without changing the total: the signature is what makes an
    // image written under one setting unloadable under another
    uint64_t sig = 1469598103934665603ull;
    void tag(uint64_t v) {
        for (int i = 0; i < 8; ++i) {
            sig ^= (v >> (8 * i)) & 0xff;
            sig *= 1099511628211ull;
        }
    }
    enum Form : uint64_t { F_CONV = 1, F_CONVT = 2, F_FRAG = 3, F_FRAG_K2 = 4, F_FRAG_K4 = 5, F_RESAMPLE = 6, F_VEC = 7, F_FRAG_BF3 = 8 };
    size_t add_untagged(size_t n) {
        size_t o = total;
        total += ((n + 3) / 4) * 4;  // keep every tensor 16-byte aligned
        if (!dry) host.resize(total, 0.0f);
        return o;
    }
    size_t add(size_t n) {
        tag(n);
        return add_untagged(n);
    }
    int layout_id(int version) const { return (int)(((sig >> 32) ^ sig ^ (uint64_t)version * 2654435761ull) & 0x7fffffff); }
    // Conv1d weight (Cout, Cin, k) -> [tap][Cout][CinP]
    size_t conv(const float* w, int cout, int cin, int k, int cinp) {
        tag(F_CONV), tag(k), tag(cout), tag(cinp);
        size_t o = add((size_t)k * cout * cinp);
        if (dry) return o;
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int t = 0; t < k; ++t) host[o + ((size_t)t * cout + co) * cinp + ci] = w[((size_t)co * cin + ci) * k + t];
        return o;
    }
    // ConvTranspose1d weight (Cin, Cout, k) -> [tap][Cout][Cin]
    size_t convT(const float* w, int cin, int cout, int k) {
        tag(F_CONVT), tag(k), tag(cout), tag(cin);
        size_t o = add((size_t)k * cout * cin);
        if (dry) return o;
        for (int ci = 0; ci < cin; ++ci)
            for (int co = 0; co < cout; ++co)
                for (int t = 0; t < k; ++t) host[o + ((size_t)t * cout + co) * cin + ci] = w[((size_t)ci * cout + co) * k + t];
        return o;
    }
    // Conv1d k5 weight (Cout, Cin, 5) [+ the block's residual 1x1 conv (Cout, Cin, 1)] -> the B-fragment stream of
    // wide_conv_kernel: [Cout/32][CinP/8][slots][64][4], slots = the taps that can be valid at length L (+ the residual)
    size_t conv_frag(const float* w, const float* wres, int cout, int cin, int cinp, int L) {
        if (bf3_select(cout, L, WK_K5)) {  // bf3.hip: stream of bf16 triples [Cout/16][CinP/32][slots][3][64][8 bf16] (counted in floats here)
            const int nslot = 5 + (wres ? 1 : 0);
            tag(F_FRAG_BF3), tag(cout), tag(cinp), tag(L), tag(wres ? 1 : 0);
            const size_t o = add(bf3_stream_elems(cout, cinp, nslot) / 2);
            if (dry) return o;
            std::vector<float> tmp((size_t)6 * cout * cinp, 0.0f);
            for (int co = 0; co < cout; ++co)
                for (int ci = 0; ci < cin; ++ci) {
                    for (int t = 0; t < 5; ++t) tmp[((size_t)t * cout + co) * cinp + ci] = w[((size_t)co * cin + ci) * 5 + t];
                    if (wres) tmp[((size_t)5 * cout + co) * cinp + ci] = wres[(size_t)co * cin + ci];
                }
            pack_fragments_bf3(tmp.data(), cout, cinp, 5, wres != nullptr, reinterpret_cast<unsigned short*>(&host[o]));
            return o;
        }
        const int sw = wide_ms(cout, L, WK_K5);
        const int kt0 = (L == 2) ? 1 : 0, ntap = (L == 2) ? 3 : 5, nslab = ntap + (wres ? 1 : 0);
        tag((L == 4 && sw == 32 && karatsuba_l4()) ? F_FRAG_K4 : (L == 2 && sw == 32 && cout / 8 == 64 && karatsuba_l2()) ? F_FRAG_K2 : F_FRAG);
        tag(cout), tag(cinp), tag(L), tag(wres ? 1 : 0), tag(sw);
        if (dry) return add((size_t)(cout / sw) * (cinp / (sw == 32 ? 8 : 16)) * ((L == 4 && sw == 32 && karatsuba_l4()) ? 9 + (wres ? 1 : 0) : nslab) * 256);
        std::vector<float> tmp((size_t)6 * cout * cinp, 0.0f);
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci) {
                for (int t = 0; t < 5; ++t) tmp[((size_t)t * cout + co) * cinp + ci] = w[((size_t)co * cin + ci) * 5 + t];
                if (wres) tmp[((size_t)5 * cout + co) * cinp + ci] = wres[(size_t)co * cin + ci];
            }
        const bool k4 = L == 4 && sw == 32 && karatsuba_l4();
        size_t o = k4 ? add_untagged((size_t)(cout / sw) * (cinp / 8) * nslab * 256) : add((size_t)(cout / sw) * (cinp / (sw == 32 ? 8 : 16)) * nslab * 256);
        if (k4) {
            // nine slots (+ residual) instead of five: the stream is longer than `o` was sized for - re-reserve (signature
            // unaffected: the dry path tags the nine-slot size only)
            total = o;
            if (!dry) host.resize(total);
            o = add((size_t)(cout / 32) * (cinp / 8) * (9 + (wres ? 1 : 0)) * 256);
            pack_fragments_k4(tmp.data(), cout, cinp, wres != nullptr, &host[o]);
        } else if (L == 2 && sw == 32 && cout / 8 == 64 && karatsuba_l2()) pack_fragments_k2(tmp.data(), cout, cinp, wres != nullptr, &host[o]);
        else pack_fragments(tmp.data(), cout, cinp, kt0, ntap, wres != nullptr, &host[o], sw);
        return o;
    }
    // strided Conv1d k3 (Cout, Cin, 3) or ConvTranspose1d k4 (Cin, Cout, 4) of a wide level -> fragment stream, slot = tap
    size_t resample_frag(const float* w, int cin, int cout, int k, bool transposed, int Lin) {
        if (bf3_select(cout, Lin, transposed ? WK_UP : WK_DOWN)) {  // bf3.hip: bf16-triple stream, slot = tap
            tag(F_FRAG_BF3), tag(F_RESAMPLE), tag(cout), tag(cin), tag(k), tag(transposed ? 1 : 0);
            const size_t o = add(bf3_stream_elems(cout, cin, k) / 2);
            if (dry) return o;
            std::vector<float> tmp((size_t)6 * cout * cin, 0.0f);
            for (int co = 0; co < cout; ++co)
                for (int ci = 0; ci < cin; ++ci)
                    for (int t = 0; t < k; ++t)
                        tmp[((size_t)t * cout + co) * cin + ci] = transposed ? w[((size_t)ci * cout + co) * k + t] : w[((size_t)co * cin + ci) * k + t];
            pack_fragments_bf3(tmp.data(), cout, cin, k, false, reinterpret_cast<unsigned short*>(&host[o]));
            return o;
        }
        const int sw = wide_ms(cout, Lin, transposed ? WK_UP : WK_DOWN);
        tag(F_RESAMPLE), tag(cout), tag(cin), tag(k), tag(transposed ? 1 : 0), tag(sw);
        if (dry) return add((size_t)(cout / sw) * (cin / (sw == 32 ? 8 : 16)) * k * 256);
        std::vector<float> tmp((size_t)6 * cout * cin, 0.0f);
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int t = 0; t < k; ++t)
                    tmp[((size_t)t * cout + co) * cin + ci] = transposed ? w[((size_t)ci * cout + co) * k + t] : w[((size_t)co * cin + ci) * k + t];
        size_t o = add((size_t)(cout / sw) * (cin / (sw == 32 ? 8 : 16)) * k * 256);
        pack_fragments(tmp.data(), cout, cin, 0, k, false, &host[o], sw);
        return o;
    }
    size_t vec(const float* v, int n) {
        tag(F_VEC);
        size_t o = add(n);
        if (!dry) memcpy(&host[o], v, sizeof(float) * n);
        return o;
    }
};

struct BufPool {
    std::vector<int> free_ids;
    std::vector<int> pinned;  // never recycled (activation taps kept readable after a forward)
    int n = 0;
    int get() {
        if (!free_ids.empty()) {
            int i = free_ids.back();
            free_ids.pop_back();
            return i;
        }
        return n++;
    }
    void pin(int i) { pinned.push_back(i); }
    void put(int i) {
        for (int q : pinned)
            if (q == i) return;
        for (int q : free_ids)
            if (q == i) return;
        free_ids.push_back(i);
    }
};

// Build-time tensor handle
struct TH {
    int buf;
    int C, L;
};

}  // namespace edmp

using namespace edmp;

extern "C" int64_t edmp_unet_param_count(const edmp_unet_desc* desc) {
    if (!desc || desc->n_levels < 2 || desc->n_levels > EDMP_MAX_LEVELS) return -1;
    return inventory(*desc).total;
}

// Version of the packing code: bump whenever the packing of any kernel family changes.  The layout id of an image is this
// version mixed with the signature of the tensor sequence the builder actually produced (Packer::sig): a stale image, or
// one written under other builder switches, fails to load instead of feeding a kernel the wrong fragment order.
static const int kPackVersion = 301;

// ---- step 1 of a model build: the LAYER PLAN -----------------------------------------------------------------------------
// An op of the plan: which kernel family, which activation buffers (ids of the pool) and which tensors of the packed weight image
// (float offsets); resolved to device pointers by resolve_program() once the image and the buffers exist.
struct POp {
    OpKind kind;
    int src1, src2, C1, C2, Lin, Lout, ntaps, stride, pad, transposed, Cout;
    size_t w, b;
    int dst;
    // gn
    int y, L, C, res;
    size_t gamma, beta;
    int tb_off;
    double fn, fe, fd;  // FLOPs per trajectory: nominal | issued | direct form without padding taps (0: same as fe)
    size_t br;  // bias of a residual 1x1 conv folded into an OP_RCB
    int blk;
    int res_out;  // OP_RCB: buffer receiving the folded residual 1x1 conv (-1: none)
    // OP_LVL: offsets of the level's tensors in the packed image, in LevelP order; lv_skip: buffer of the skip output (-1: none)
    size_t lvo[24];
    int lv_variant, lv_tb1, lv_tb2, lv_skip, lv_merge;
    // fused conv+gn (OP_RCB): uses src1/src2/C1/C2/Lin/Cout/w/b/dst + gamma/beta/res/tb_off
};

// the builder's run-time switches, read ONCE per model build and frozen into its plan (two models built under different settings
// coexist in one process: A/B runs, the adversarial-weights test)
struct BuildSwitches {
    bool fused, resfold, level;
    static BuildSwitches read() {
        BuildSwitches s;
        s.fused = getenv("EDMP_NO_FUSED") == nullptr;
        s.resfold = getenv("EDMP_NO_RESFOLD") == nullptr;
        s.level = s.fused && getenv("EDMP_NO_LEVEL") == nullptr;
        return s;
    }
};

// Walks the architecture (temporalunet.py:47-76) once: decides per layer which kernel family runs it, packs its weights into the
// device image in that family's layout (or only sizes them: Packer::dry, loading a packed image), assigns activation buffers from a
// recycling pool and records the FLOP counts.  No device work.
struct LayerPlan {
    const edmp_unet_desc* desc;
    const float* params;   // state-dict blob (never read through when pk.dry)
    RawNet inv;
    BuildSwitches sw;
    int N, td, nd, CP0 = 8;  // horizon, time_dim, levels, padded input channels
    std::vector<int> dm;     // input_dim, dims...
    // products
    Packer pk;
    BufPool pool;
    std::vector<POp> pops;
    std::vector<float> tw_all, tb_all;  // concatenated time-MLP weights of the residual blocks
    int tb_cursor = 0, rcb_idx = 0;
    int x_in_buf = -1, head_buf = -1;
    bool final_fused = false;
    struct TapRec { int which; int buf, C, L; };
    std::vector<TapRec> tapr;
    size_t hw = 0, hb = 0, o_t1w = 0, o_t1b = 0, o_t3w = 0, o_t3b = 0, o_tw = 0, o_tb = 0;
    double head_flops = 0;

    LayerPlan(const edmp_unet_desc* d, const float* blob, bool layout_only) : desc(d), params(blob), inv(inventory(*d)), sw(BuildSwitches::read()) {
        N = d->horizon, td = d->time_dim, nd = d->n_levels;
        dm.push_back(d->input_dim);
        for (int i = 0; i < d->n_levels; ++i) dm.push_back(d->dims[i]);
        pk.dry = layout_only;
    }


        void append(std::vector<float>& v, const float* src, size_t n) {  // layout-only pass: sizes, no reads
            if (pk.dry) v.resize(v.size() + n, 0.0f);
            else v.insert(v.end(), src, src + n);
        }

        static long valid_pairs(int Lin, int Lout, int k, int stride, int pad, bool tr) {
            long cnt = 0;
            for (int lo = 0; lo < Lout; ++lo)
                for (int t = 0; t < k; ++t) {
                    if (!tr) {
                        int li = lo * stride + t - pad;
                        if (li >= 0 && li < Lin) ++cnt;
                    } else {
                        int num = lo + pad - t;
                        if (num >= 0 && num % stride == 0 && num / stride < Lin) ++cnt;
                    }
                }
            return cnt;
        }
        TH emit_conv(TH a, const TH* a2, int cin_true, size_t w, size_t b, int Cout, int k, int stride, int pad, bool tr, int Lout) {
            POp o{};
            o.kind = OP_CONV;
            o.src1 = a.buf;
            o.C1 = a.C;
            o.src2 = a2 ? a2->buf : -1;
            o.C2 = a2 ? a2->C : 0;
            o.Lin = a.L;
            o.Lout = Lout;
            o.ntaps = k;
            o.stride = stride;
            o.pad = pad;
            o.transposed = tr;
            o.Cout = Cout;
            o.w = w;
            o.b = b;
            o.dst = pool.get();
            // nominal FLOPs: what torch executes: conv 2*Lout*Cout*Cin*k ; convT 2*Lin*Cin*Cout*k (uncropped)
            o.fn = tr ? 2.0 * a.L * cin_true * Cout * k : 2.0 * Lout * Cout * (double)cin_true * k;
            o.fe = 2.0 * (double)valid_pairs(a.L, Lout, k, stride, pad, tr) * Cout * (double)(o.C1 + o.C2);
            pops.push_back(o);
            return TH{o.dst, Cout, Lout};
        }
        TH emit_wrs(TH a, size_t w, size_t b, int Cout, int kind, int k, int Lout) {
            POp o{};
            o.kind = OP_WRS;
            o.src1 = a.buf;
            o.C1 = a.C;
            o.src2 = -1;
            o.C2 = 0;
            o.Lin = a.L;
            o.Lout = Lout;
            o.ntaps = k;
            o.transposed = (kind == WK_UP);
            o.Cout = Cout;
            o.w = w;
            o.b = b;
            o.dst = pool.get();
            o.res_out = -1;
            o.blk = kind;
            const bool tr = kind == WK_UP;
            o.fn = tr ? 2.0 * a.L * (double)a.C * Cout * k : 2.0 * Lout * Cout * (double)a.C * k;
            o.fe = 2.0 * (double)valid_pairs(a.L, Lout, k, 2, 1, tr) * Cout * (double)a.C;
            pops.push_back(o);
            return TH{o.dst, Cout, Lout};
        }
        void emit_gn(TH y, size_t gamma, size_t beta, int res_buf, int tb_off) {
            POp o{};
            o.kind = OP_GN;
            o.y = y.buf;
            o.L = y.L;
            o.C = y.C;
            o.gamma = gamma;
            o.beta = beta;
            o.res = res_buf;
            o.tb_off = tb_off;
            pops.push_back(o);
        }
        TH emit_fused(TH a, const TH* a2, int cin_true, int cout, size_t w, size_t b, size_t gamma, size_t beta, int res_buf, int tbo) {
            POp o{};
            o.kind = OP_RCB;
            o.src1 = a.buf;
            o.C1 = a.C;
            o.src2 = a2 ? a2->buf : -1;
            o.C2 = a2 ? a2->C : 0;
            o.Lin = a.L;
            o.Lout = a.L;
            o.ntaps = 5;
            o.Cout = cout;
            o.w = w;
            o.b = b;
            o.gamma = gamma;
            o.beta = beta;
            o.res = res_buf;
            o.tb_off = tbo;
            o.res_out = -1;
            o.dst = pool.get();
            o.fn = 2.0 * a.L * cout * (double)cin_true * 5;
            // executed = issued MFMA work: the L = 2 Karatsuba form runs 3 matrix products where the direct form runs 4
            const bool k2 = a.L == 2 && rcb_form(cout, a.L) == 2 && rcb_supported(cout, a.L, o.C1, o.C2);
            const bool k4 = a.L == 4 && rcb_form(cout, a.L) == 4 && rcb_supported(cout, a.L, o.C1, o.C2);
            o.fe = 2.0 * (k2 ? 3.0 : k4 ? 9.0 : (double)valid_pairs(a.L, a.L, 5, 1, 2, false)) * cout * (double)(o.C1 + o.C2);
            o.fd = 2.0 * (double)valid_pairs(a.L, a.L, 5, 1, 2, false) * cout * (double)(o.C1 + o.C2);
            pops.push_back(o);
            return TH{o.dst, cout, a.L};
        }
        TH emit_rcb(TH x, const TH* x2) {
            const RawRCB& r = inv.rcbs[rcb_idx++];
            const int cin_store = x.C + (x2 ? x2->C : 0);
            // conv1 (a residual 1x1 conv folded into the wide fused kernel is packed right behind it, as tap index 5)
            const bool wide = sw.fused && rcb_supported(r.cout, x.L, x.C, x2 ? x2->C : 0);
            const bool fold_res = wide && r.has_res && sw.resfold;
            // the position-tile kernel reads its weights as an MFMA fragment stream (wide.hip); the generic conv as [tap][Cout][Cin]
            size_t w1 = wide ? pk.conv_frag(params + r.cb[0].w.off, fold_res ? params + r.rw.off : nullptr, r.cout, r.cin, cin_store, x.L)
                             : pk.conv(params + r.cb[0].w.off, r.cout, r.cin, 5, cin_store);
            size_t b1 = pk.vec(params + r.cb[0].b.off, r.cout);
            size_t g1 = pk.vec(params + r.cb[0].gw.off, r.cout), be1 = pk.vec(params + r.cb[0].gb.off, r.cout);
            size_t w2 = wide ? pk.conv_frag(params + r.cb[1].w.off, nullptr, r.cout, r.cout, r.cout, x.L) : pk.conv(params + r.cb[1].w.off, r.cout, r.cout, 5, r.cout);
            size_t b2 = pk.vec(params + r.cb[1].b.off, r.cout);
            size_t g2 = pk.vec(params + r.cb[1].gw.off, r.cout), be2 = pk.vec(params + r.cb[1].gb.off, r.cout);
            int tb_off = tb_cursor;
            tb_cursor += r.cout;
            append(tw_all, params + r.tw.off, (size_t)r.cout * td);
            append(tb_all, params + r.tb.off, r.cout);
            if (wide) {
                TH h = emit_fused(x, x2, r.cin, r.cout, w1, b1, g1, be1, -1, tb_off);
                int res_buf;
                int rr_buf = -1;
                if (fold_res) {
                    POp& c1 = pops.back();
                    c1.res_out = pool.get();
                    c1.br = pk.vec(params + r.rb.off, r.cout);
                    c1.fn += 2.0 * x.L * r.cout * (double)r.cin;
                    c1.fe += 2.0 * x.L * r.cout * (double)cin_store;
                    c1.fd += 2.0 * x.L * r.cout * (double)cin_store;
                    rr_buf = c1.res_out;
                    res_buf = c1.res_out;
                } else if (r.has_res) {
                    // (EDMP_NO_RESFOLD: the residual 1x1 conv as its own launch)
                    size_t wr = pk.conv(params + r.rw.off, r.cout, r.cin, 1, cin_store);
                    size_t br = pk.vec(params + r.rb.off, r.cout);
                    TH rr = emit_conv(x, x2, r.cin, wr, br, r.cout, 1, 1, 0, false, x.L);
                    rr_buf = rr.buf;
                    res_buf = rr.buf;
                } else {
                    res_buf = x2 ? -2 : x.buf;
                }
                TH out = emit_fused(h, nullptr, r.cout, r.cout, w2, b2, g2, be2, res_buf, -1);
                pool.put(h.buf);
                if (rr_buf >= 0) pool.put(rr_buf);
                return out;
            }
            TH y1 = emit_conv(x, x2, r.cin, w1, b1, r.cout, 5, 1, 2, false, x.L);
            emit_gn(y1, g1, be1, -1, tb_off);
            TH y2 = emit_conv(y1, nullptr, r.cout, w2, b2, r.cout, 5, 1, 2, false, x.L);
            pool.put(y1.buf);
            if (r.has_res) {
                size_t wr = pk.conv(params + r.rw.off, r.cout, r.cin, 1, cin_store);
                size_t br = pk.vec(params + r.rb.off, r.cout);
                TH rr = emit_conv(x, x2, r.cin, wr, br, r.cout, 1, 1, 0, false, x.L);
                emit_gn(y2, g2, be2, rr.buf, -1);
                pool.put(rr.buf);
            } else {
                emit_gn(y2, g2, be2, x2 ? -2 : x.buf, -1);  // identity residual (blocks.py:151-152); -2 = unsupported concat
            }
            return y2;
        }

        // a whole level in one launch (level.hip): RCB, RCB, resampling conv (+ the final Conv1dBlock); consumes two entries
        // of inv.rcbs like two emit_rcb calls would, in the same order (so the time-bias row keeps its layout)
        TH emit_level(int mode, int variant, TH xin, const TH* x2, const RawT& rs_w, const RawT& rs_b, bool want_skip, TH* skip_th) {
            const RawRCB& r1 = inv.rcbs[rcb_idx++];
            const RawRCB& r2 = inv.rcbs[rcb_idx++];
            const int Cc = r1.cout, Ll = xin.L, KX = level_kx(variant);
            const int cin_store = xin.C + (x2 ? x2->C : 0);
            POp o{};
            o.kind = OP_LVL;
            o.lv_variant = variant;
            o.src1 = xin.buf;
            o.C1 = xin.C;
            o.src2 = x2 ? x2->buf : -1;
            o.C2 = x2 ? x2->C : 0;
            o.Lin = Ll;
            o.Cout = Cc;
            size_t* q = o.lvo;
            // LevelP order: w11 w12 w21 w22 wrs wfin | b11 g11 be11 rb1 | b12 g12 be12 | b21 g21 be21 | b22 g22 be22 | brs | bfin gfin befin
            q[0] = pk.conv_frag(params + r1.cb[0].w.off, params + r1.rw.off, Cc, r1.cin, KX, Ll);
            q[1] = pk.conv_frag(params + r1.cb[1].w.off, nullptr, Cc, Cc, Cc, Ll);
            q[2] = pk.conv_frag(params + r2.cb[0].w.off, nullptr, Cc, Cc, Cc, Ll);
            q[3] = pk.conv_frag(params + r2.cb[1].w.off, nullptr, Cc, Cc, Cc, Ll);
            q[4] = pk.resample_frag(params + rs_w.off, Cc, Cc, mode == LV_DOWN ? 3 : 4, mode != LV_DOWN, Ll);
            q[5] = (mode == LV_UP_FINAL) ? pk.conv_frag(params + inv.final_cb.w.off, nullptr, Cc, Cc, Cc, 50) : 0;
            q[6] = pk.vec(params + r1.cb[0].b.off, Cc), q[7] = pk.vec(params + r1.cb[0].gw.off, Cc), q[8] = pk.vec(params + r1.cb[0].gb.off, Cc);
            q[9] = pk.vec(params + r1.rb.off, Cc);
            q[10] = pk.vec(params + r1.cb[1].b.off, Cc), q[11] = pk.vec(params + r1.cb[1].gw.off, Cc), q[12] = pk.vec(params + r1.cb[1].gb.off, Cc);
            q[13] = pk.vec(params + r2.cb[0].b.off, Cc), q[14] = pk.vec(params + r2.cb[0].gw.off, Cc), q[15] = pk.vec(params + r2.cb[0].gb.off, Cc);
            q[16] = pk.vec(params + r2.cb[1].b.off, Cc), q[17] = pk.vec(params + r2.cb[1].gw.off, Cc), q[18] = pk.vec(params + r2.cb[1].gb.off, Cc);
            q[19] = pk.vec(params + rs_b.off, Cc);
            if (mode == LV_UP_FINAL) {
                q[20] = pk.vec(params + inv.final_cb.b.off, Cc), q[21] = pk.vec(params + inv.final_cb.gw.off, Cc), q[22] = pk.vec(params + inv.final_cb.gb.off, Cc);
            }
            for (const RawRCB* r : {&r1, &r2}) {  // time-bias table columns, block order
                (r == &r1 ? o.lv_tb1 : o.lv_tb2) = tb_cursor;
                tb_cursor += Cc;
                append(tw_all, params + r->tw.off, (size_t)Cc * td);
                append(tb_all, params + r->tb.off, Cc);
            }
            o.lv_skip = -1;
            if (want_skip) {
                o.lv_skip = pool.get();
                *skip_th = TH{o.lv_skip, Cc, Ll};
            }
            int Lout = (mode == LV_DOWN) ? (Ll - 1) / 2 + 1 : 2 * Ll;
            if (mode != LV_DOWN && (Lout == 8 || Lout == 14 || Lout == 26)) Lout -= 1;
            o.Lout = Lout;
            o.dst = pool.get();
            const double vp = (double)valid_pairs(Ll, Ll, 5, 1, 2, false);
            const int k = mode == LV_DOWN ? 3 : 4;
            o.fn = 2.0 * Ll * Cc * 5.0 * ((double)r1.cin + 3.0 * Cc) + 2.0 * Ll * Cc * (double)r1.cin +
                   (mode == LV_DOWN ? 2.0 * Lout * Cc * (double)Cc * k : 2.0 * Ll * (double)Cc * Cc * k) + (mode == LV_UP_FINAL ? 2.0 * Lout * Cc * (double)Cc * 5 : 0.0);
            o.fe = 2.0 * vp * Cc * ((double)cin_store + 3.0 * Cc) + 2.0 * Ll * Cc * (double)cin_store +
                   2.0 * (double)valid_pairs(Ll, Lout, k, 2, 1, mode != LV_DOWN) * Cc * (double)Cc +
                   (mode == LV_UP_FINAL ? 2.0 * (double)valid_pairs(Lout, Lout, 5, 1, 2, false) * Cc * (double)Cc : 0.0);
            pops.push_back(o);
            return TH{o.dst, Cc, Lout};
        }

    // the walk itself: down path, middle, up path, final block + head, time-embedding tensors
    int plan() {

        TH x{pool.get(), CP0, N};
        x_in_buf = x.buf;
        pool.pin(x_in_buf);  // written by the sampler kernels between forwards: never recycled as an activation
        std::vector<TH> skips;
        for (int i = 0; i < nd; ++i) {
            if (const int lvv = (sw.level && i != nd - 1) ? level_variant(LV_DOWN, dm[i + 1], x.L, x.C, 0) : 0) {
                // the skip of level 0 is never consumed (5 up-samplers for 6 skips, temporalunet.py:31-32,66-67): not even written
                TH sk{-1, dm[i + 1], x.L};
                TH xo = emit_level(LV_DOWN, lvv, x, nullptr, inv.down_w[i], inv.down_b[i], i > 0, &sk);
                pool.put(x.buf);
                skips.push_back(sk);
                x = xo;
                // merged with the next down level (level2_kernel): this level's output only ever exists in LDS - no tap
                const bool merged = lvv == 1 && (level_merge_mask() & 1) && i + 1 < nd - 1 && level_variant(LV_DOWN, dm[i + 2], x.L, x.C, 0) == 2;
                pops.back().lv_merge = merged ? 1 : 0;
                if (!merged) {
                    tapr.push_back({i, x.buf, x.C, x.L});
                    pool.pin(x.buf);
                }
                continue;
            }
            TH a = emit_rcb(x, nullptr);
            pool.put(x.buf);  // the block input is dead once both consumers (conv1, residual) are emitted
            TH b = emit_rcb(a, nullptr);
            pool.put(a.buf);
            skips.push_back(b);
            if (i != nd - 1) {
                int Lout = (b.L - 1) / 2 + 1;
                if (sw.fused && wrs_supported(dm[i + 1], b.C, b.L, false)) {
                    size_t w = pk.resample_frag(params + inv.down_w[i].off, dm[i + 1], dm[i + 1], 3, false, b.L);
                    size_t bb = pk.vec(params + inv.down_b[i].off, dm[i + 1]);
                    x = emit_wrs(b, w, bb, dm[i + 1], WK_DOWN, 3, Lout);
                } else {
                    size_t w = pk.conv(params + inv.down_w[i].off, dm[i + 1], dm[i + 1], 3, dm[i + 1]);
                    size_t bb = pk.vec(params + inv.down_b[i].off, dm[i + 1]);
                    x = emit_conv(b, nullptr, dm[i + 1], w, bb, dm[i + 1], 3, 2, 1, false, Lout);
                }
            } else {
                x = b;
            }
            tapr.push_back({i, x.buf, x.C, x.L});
            pool.pin(x.buf);
        }
        {
            // middle: input is skips.back() (same buffer, must stay alive for the up path)
            TH a = emit_rcb(x, nullptr);
            TH b = emit_rcb(a, nullptr);
            pool.put(a.buf);
            x = b;
            tapr.push_back({100, x.buf, x.C, x.L});
            pool.pin(x.buf);
        }
        for (int j = 0, i = nd; i > 1; --i, ++j) {
            TH sk = skips.back();
            skips.pop_back();
            EDMP_REQUIRE(sk.L == x.L && sk.C == x.C, "skip/upsample shape mismatch at up level %d (L %d vs %d)", j, sk.L, x.L);
            EDMP_REQUIRE(sk.buf >= 0, "up level %d consumes a skip tensor that the fused down level did not write", j);
            {
                const bool last = (i == 2);
                const int mode = (last && dm[1] == dm[i - 1] && 2 * x.L == N) ? LV_UP_FINAL : LV_UP;
                if (const int lvv = sw.level ? level_variant(mode, dm[i - 1], x.L, x.C, sk.C) : 0) {
                    TH xo = emit_level(mode, lvv, x, &sk, inv.up_w[j], inv.up_b[j], false, nullptr);
                    pool.put(x.buf);
                    pool.put(sk.buf);
                    x = xo;
                    if (mode == LV_UP_FINAL) {
                        final_fused = true;  // the level kernel already applied final_conv.0
                    } else {
                        // merged with the last up level (level2_kernel): this level's output only ever exists in LDS - no tap
                        const bool merged = lvv == 3 && (level_merge_mask() & 2) && i == 3 && !skips.empty() && dm[1] == dm[i - 2] && 2 * x.L == N &&
                                            level_variant(LV_UP_FINAL, dm[i - 2], x.L, x.C, skips.back().C) == 4;
                        pops.back().lv_merge = merged ? 1 : 0;
                        if (!merged) {
                            tapr.push_back({200 + j, x.buf, x.C, x.L});
                            pool.pin(x.buf);
                        }
                    }
                    continue;
                }
            }
            TH a = emit_rcb(x, &sk);
            pool.put(x.buf);
            pool.put(sk.buf);
            TH b = emit_rcb(a, nullptr);
            pool.put(a.buf);
            int Lout = 2 * b.L;
            if (Lout == 8 || Lout == 14 || Lout == 26) Lout -= 1;  // crop rule, temporalunet.py:70-71
            if (sw.fused && wrs_supported(dm[i - 1], b.C, b.L, true)) {
                size_t w = pk.resample_frag(params + inv.up_w[j].off, dm[i - 1], dm[i - 1], 4, true, b.L);
                size_t bb = pk.vec(params + inv.up_b[j].off, dm[i - 1]);
                x = emit_wrs(b, w, bb, dm[i - 1], WK_UP, 4, Lout);
            } else {
                size_t w = pk.convT(params + inv.up_w[j].off, dm[i - 1], dm[i - 1], 4);
                size_t bb = pk.vec(params + inv.up_b[j].off, dm[i - 1]);
                x = emit_conv(b, nullptr, dm[i - 1], w, bb, dm[i - 1], 4, 2, 1, true, Lout);
            }
            pool.put(b.buf);
            tapr.push_back({200 + j, x.buf, x.C, x.L});
            pool.pin(x.buf);
        }
        EDMP_REQUIRE(x.L == N, "decoder output length %d != horizon %d", x.L, N);
        // final Conv1dBlock + 1x1 head
        if (!final_fused) {
            size_t w = pk.conv(params + inv.final_cb.w.off, dm[1], dm[1], 5, dm[1]);
            size_t b = pk.vec(params + inv.final_cb.b.off, dm[1]);
            size_t g = pk.vec(params + inv.final_cb.gw.off, dm[1]), be = pk.vec(params + inv.final_cb.gb.off, dm[1]);
            TH y = emit_conv(x, nullptr, dm[1], w, b, dm[1], 5, 1, 2, false, N);
            emit_gn(y, g, be, -1, -1);
            pool.put(x.buf);
            x = y;
        }
        hw = pk.vec(params + inv.final_w.off, desc->input_dim * dm[1]);
        hb = pk.vec(params + inv.final_b.off, desc->input_dim);
        head_flops = 2.0 * N * desc->input_dim * dm[1];

        // raw time-embedding weights for the table kernel
        o_t1w = pk.vec(params + inv.t1w.off, 4 * td * td), o_t1b = pk.vec(params + inv.t1b.off, 4 * td);
        o_t3w = pk.vec(params + inv.t3w.off, 4 * td * td), o_t3b = pk.vec(params + inv.t3b.off, td);
        o_tw = pk.vec(tw_all.data(), (int)tw_all.size()), o_tb = pk.vec(tb_all.data(), (int)tb_all.size());
        head_buf = x.buf;
        for (auto& o : pops) EDMP_REQUIRE(!((o.kind == OP_GN || o.kind == OP_RCB) && o.res == -2), "identity residual over a channel concat is not supported");
        return EDMP_OK;
    }
};

// ---- step 3: buffer ids / image offsets of the plan -> device pointers of the loaded model (the layer program unet_run_program walks)
static void resolve_program(UNet* u, const LayerPlan& pl) {
    const std::vector<POp>& pops = pl.pops;
    for (auto& o : pops) {
        Op op{};
        op.kind = o.kind;
        op.tb_off = -1;
        if (o.kind == OP_CONV) {
            ConvP& c = op.cv;
            c.src1 = u->bufs[o.src1];
            c.src2 = o.src2 >= 0 ? u->bufs[o.src2] : nullptr;
            c.C1 = o.C1;
            c.C2 = o.C2;
            c.Lin = o.Lin;
            c.Lout = o.Lout;
            c.ntaps = o.ntaps;
            c.stride = o.stride;
            c.pad = o.pad;
            c.transposed = o.transposed;
            c.W = u->wpack + o.w;
            c.bias = u->wpack + o.b;
            c.dst = u->bufs[o.dst];
            c.Cout = o.Cout;
            op.flops_nominal = o.fn;
            op.flops_exec = o.fe;
            u->flops_nominal += o.fn;
            u->flops_exec += o.fe;
            op.flops_direct = o.fd > 0 ? o.fd : o.fe;
            u->flops_direct += op.flops_direct;
        } else if (o.kind == OP_LVL) {
            LevelP& c = op.lv;
            const float* W0 = u->wpack;
            c.src1 = u->bufs[o.src1];
            c.src2 = o.src2 >= 0 ? u->bufs[o.src2] : nullptr;
            c.C1 = o.C1;
            c.C2 = o.C2;
            c.w11 = W0 + o.lvo[0], c.w12 = W0 + o.lvo[1], c.w21 = W0 + o.lvo[2], c.w22 = W0 + o.lvo[3], c.wrs = W0 + o.lvo[4], c.wfin = W0 + o.lvo[5];
            c.b11 = W0 + o.lvo[6], c.g11 = W0 + o.lvo[7], c.be11 = W0 + o.lvo[8], c.rb1 = W0 + o.lvo[9];
            c.b12 = W0 + o.lvo[10], c.g12 = W0 + o.lvo[11], c.be12 = W0 + o.lvo[12];
            c.b21 = W0 + o.lvo[13], c.g21 = W0 + o.lvo[14], c.be21 = W0 + o.lvo[15];
            c.b22 = W0 + o.lvo[16], c.g22 = W0 + o.lvo[17], c.be22 = W0 + o.lvo[18];
            c.brs = W0 + o.lvo[19];
            c.bfin = W0 + o.lvo[20], c.gfin = W0 + o.lvo[21], c.befin = W0 + o.lvo[22];
            c.tb1 = c.tb2 = nullptr;
            c.skip_out = o.lv_skip >= 0 ? u->bufs[o.lv_skip] : nullptr;
            c.out = u->bufs[o.dst];
            op.lv_variant = o.lv_variant;
            op.lv_merge = o.lv_merge;
            op.lv_sb = level_sb(o.lv_variant);
            op.lv_tb1 = o.lv_tb1;
            op.lv_tb2 = o.lv_tb2;
            op.flops_nominal = o.fn;
            op.flops_exec = o.fe;
            u->flops_nominal += o.fn;
            u->flops_exec += o.fe;
            op.flops_direct = o.fd > 0 ? o.fd : o.fe;
            u->flops_direct += op.flops_direct;
        } else if (o.kind == OP_WRS) {
            RcbP& c = op.rc;
            c.src1 = u->bufs[o.src1];
            c.src2 = nullptr;
            c.C1 = o.C1;
            c.C2 = 0;
            c.W = u->wpack + o.w;
            c.bias = u->wpack + o.b;
            c.dst = u->bufs[o.dst];
            c.Cout = o.Cout;
            op.rc_L = o.Lin;
            op.wrs_kind = o.blk;
            op.rc_ms = wide_ms(o.Cout, o.Lin, o.blk);
            op.rc_bf3 = bf3_select(o.Cout, o.Lin, o.blk) ? 1 : 0;
            op.flops_nominal = o.fn;
            op.flops_exec = o.fe;
            u->flops_nominal += o.fn;
            u->flops_exec += o.fe;
            op.flops_direct = o.fd > 0 ? o.fd : o.fe;
            u->flops_direct += op.flops_direct;
            if (op.rc_bf3) {
                op.flops_bf16 = 6.0 * op.flops_direct;
                u->flops_bf16 += op.flops_bf16;
                u->flops_f32_moved += op.flops_exec;
            }
        } else if (o.kind == OP_RCB) {
            RcbP& c = op.rc;
            c.src1 = u->bufs[o.src1];
            c.src2 = o.src2 >= 0 ? u->bufs[o.src2] : nullptr;
            c.C1 = o.C1;
            c.C2 = o.C2;
            c.W = u->wpack + o.w;
            c.bias = u->wpack + o.b;
            c.res_out = o.res_out >= 0 ? u->bufs[o.res_out] : nullptr;
            c.res_bias = o.res_out >= 0 ? u->wpack + o.br : nullptr;
            c.gamma = u->wpack + o.gamma;
            c.beta = u->wpack + o.beta;
            c.add_tb = nullptr;
            c.add_res = o.res >= 0 ? u->bufs[o.res] : nullptr;
            c.dst = u->bufs[o.dst];
            c.Cout = o.Cout;
            op.rc_L = o.Lin;
            op.rc_form = rcb_form(o.Cout, o.Lin);
            op.rc_ms = wide_ms(o.Cout, o.Lin, WK_K5);
            op.rc_bf3 = (op.rc_form == 0 && bf3_select(o.Cout, o.Lin, WK_K5)) ? 1 : 0;
            op.tb_off = o.tb_off;
            op.flops_nominal = o.fn;
            op.flops_exec = o.fe;
            u->flops_nominal += o.fn;
            u->flops_exec += o.fe;
            op.flops_direct = o.fd > 0 ? o.fd : o.fe;
            u->flops_direct += op.flops_direct;
            if (op.rc_bf3) {  // six exact partial products per fp32 product, issued on the bf16 pipe
                op.flops_bf16 = 6.0 * op.flops_direct;
                u->flops_bf16 += op.flops_bf16;
                u->flops_f32_moved += op.flops_exec;
            }
        } else {
            GnP& g = op.gn;
            g.y = u->bufs[o.y];
            g.gamma = u->wpack + o.gamma;
            g.beta = u->wpack + o.beta;
            g.add_res = o.res >= 0 ? u->bufs[o.res] : nullptr;
            g.add_tbias = nullptr;
            g.L = o.L;
            g.C = o.C;
            op.tb_off = o.tb_off;
        }
        op_kernel_name(op, op.name);
        u->prog.push_back(op);
    }
    // a merged pair is ONE launch, attributed to its first op: that op carries the pair's FLOPs (the model totals are unaffected)
    for (size_t i = 0; i + 1 < u->prog.size(); ++i)
        if (u->prog[i].kind == OP_LVL && u->prog[i].lv_merge) {
            Op &a = u->prog[i], &b = u->prog[i + 1];
            a.flops_nominal += b.flops_nominal, a.flops_exec += b.flops_exec, a.flops_direct += b.flops_direct;
            b.flops_nominal = b.flops_exec = b.flops_direct = 0.0;
        }
}

// builds the layer program + device weight image.  packed == nullptr: repack `params` (state-dict order) on the host;
// otherwise `packed` IS the device image (edmp_unet_read_packed of the same architecture): only the layout is computed
static int unet_build(edmp_ctx* ctx, const edmp_unet_desc* desc, const float* params, int64_t n_params, int max_batch, const float* packed,
                      int64_t n_packed, int packed_layout) {
    ctx->epoch++;
    EDMP_REQUIRE(desc->n_levels >= 2 && desc->n_levels <= EDMP_MAX_LEVELS, "n_levels out of range");
    EDMP_REQUIRE(desc->input_dim >= 1 && desc->input_dim <= 8, "input_dim must be in 1..8");
    EDMP_REQUIRE(desc->time_dim >= 4 && desc->time_dim % 2 == 0, "time_dim must be even");
    EDMP_REQUIRE(max_batch >= 1, "max_batch must be positive");
    for (int i = 0; i < desc->n_levels; ++i) EDMP_REQUIRE(desc->dims[i] % 8 == 0 && desc->dims[i] >= 8, "dims must be multiples of 8");
    RawNet inv = inventory(*desc);
    static const float no_params = 0.0f;
    if (packed) {  // layout-only pass: the builder computes offsets from `params` but never reads through it (Packer::dry)
        params = &no_params;
        n_params = inv.total;
    }
    EDMP_REQUIRE(inv.total == n_params, "parameter blob has %lld floats, architecture needs %lld", (long long)n_params, (long long)inv.total);
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    if (ctx->unet) {
        unet_destroy(ctx->unet);
        ctx->unet = nullptr;
    }
    // owned here until the build has succeeded: every early return below (argument checks, failed allocations / copies /
    // launches) releases the weight image and the activation buffers
    std::unique_ptr<UNet, void (*)(UNet*)> guard(new UNet(), unet_destroy);
    UNet* u = guard.get();
    u->desc = *desc;
    u->max_batch = max_batch;
    LayerPlan pl(desc, params, packed != nullptr);
    if (int rc = pl.plan()) return rc;
    u->tb_stride = pl.tb_cursor;
    // ---- step 2: device image, activation buffers, time-bias table
    size_t max_lc = (size_t)pl.N * pl.CP0;
    for (auto& o : pl.pops)
        if (o.kind == OP_CONV || o.kind == OP_RCB || o.kind == OP_WRS || o.kind == OP_LVL) max_lc = std::max(max_lc, (size_t)std::max(o.Lout, o.Lin) * o.Cout);
    u->buf_cap = max_lc * (size_t)max_batch;
    u->layout = pl.pk.layout_id(kPackVersion);
    if (packed && packed_layout != u->layout) {
        set_error("packed weight image has layout id %d, this library with the current builder switches packs %d: re-pack from the state dict", packed_layout, u->layout);
        return EDMP_ERR_LAYOUT;
    }
    if (packed && (int64_t)pl.pk.total != n_packed) {
        set_error("packed weight image has %lld floats, this architecture / library layout needs %zu", (long long)n_packed, pl.pk.total);
        return EDMP_ERR_LAYOUT;
    }
    if (hipMalloc((void**)&u->wpack, pl.pk.total * sizeof(float)) != hipSuccess) {
        set_error("hipMalloc of %zu weight bytes failed", pl.pk.total * sizeof(float));
        return EDMP_ERR_HIP;
    }
    u->wpack_floats = pl.pk.total;
    EDMP_HIP_CHECK(hipMemcpy(u->wpack, packed ? packed : pl.pk.host.data(), pl.pk.total * sizeof(float), hipMemcpyHostToDevice));
    for (int i = 0; i < pl.pool.n; ++i) {
        float* p = nullptr;
        if (hipMalloc((void**)&p, u->buf_cap * sizeof(float)) != hipSuccess) {
            set_error("hipMalloc of activation buffer %d (%zu bytes) failed", i, u->buf_cap * sizeof(float));
            return EDMP_ERR_HIP;
        }
        u->bufs.push_back(p);
    }
    EDMP_HIP_CHECK(hipMalloc((void**)&u->tbias, (size_t)desc->T * u->tb_stride * sizeof(float)));
    {
        size_t sm = (size_t)(6 * pl.td) * sizeof(float);
        hipLaunchKernelGGL(time_table_kernel, dim3(desc->T), dim3(256), sm, ctx->stream, u->wpack + pl.o_t1w, u->wpack + pl.o_t1b,
                           u->wpack + pl.o_t3w, u->wpack + pl.o_t3b, u->wpack + pl.o_tw, u->wpack + pl.o_tb, u->tbias, pl.td, u->tb_stride);
        EDMP_HIP_CHECK(hipGetLastError());
        EDMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    resolve_program(u, pl);
    u->flops_nominal += pl.head_flops;
    u->flops_exec += pl.head_flops;
    u->flops_direct += pl.head_flops;
    u->x_in = u->bufs[pl.x_in_buf];
    u->h_last = u->bufs[pl.head_buf];
    u->head_w = u->wpack + pl.hw;
    u->head_b = u->wpack + pl.hb;
    u->head_cin = pl.dm[1];
    for (auto& t : pl.tapr) u->taps.push_back({t.which, u->bufs[t.buf], t.C, t.L});
    ctx->unet = guard.release();
    return EDMP_OK;
}

extern "C" int edmp_unet_load(edmp_ctx* ctx, const edmp_unet_desc* desc, const float* params, int64_t n_params, int max_batch) {
    EDMP_REQUIRE(ctx && desc && params, "edmp_unet_load: null argument");
    return unet_build(ctx, desc, params, n_params, max_batch, nullptr, 0, 0);
}

extern "C" int edmp_unet_load_packed(edmp_ctx* ctx, const edmp_unet_desc* desc, const float* packed, int64_t n_packed, int layout, int max_batch) {
    EDMP_REQUIRE(ctx && desc && packed, "edmp_unet_load_packed: null argument");
    return unet_build(ctx, desc, nullptr, 0, max_batch, packed, n_packed, layout);
}

extern "C" int64_t edmp_unet_packed_size(edmp_ctx* ctx, int* layout) {
    if (layout) *layout = (ctx && ctx->unet) ? ctx->unet->layout : 0;
    return (ctx && ctx->unet) ? (int64_t)ctx->unet->wpack_floats : -1;
}

extern "C" int edmp_unet_read_packed(edmp_ctx* ctx, float* out_host, int64_t capacity) {
    EDMP_REQUIRE(ctx && ctx->unet && out_host, "edmp_unet_read_packed: null argument / no model");
    EDMP_REQUIRE(capacity >= (int64_t)ctx->unet->wpack_floats, "buffer of %lld floats, image has %zu", (long long)capacity, ctx->unet->wpack_floats);
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    EDMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    EDMP_HIP_CHECK(hipMemcpy(out_host, ctx->unet->wpack, ctx->unet->wpack_floats * sizeof(float), hipMemcpyDeviceToHost));
    return EDMP_OK;
}

namespace edmp {
// shared with sampler.hip: run the forward on the context's stream
// run the layer program on u->x_in ([B][N][8], already filled); leaves the head input in u->h_last
// `tail` (device-resident loop): if the program ends with the fused final level (LV_UP_FINAL, 32 channels into the head) the
// tail of the reverse step runs inside that launch and *tail_done is set; otherwise the caller launches head_psample_kernel
int unet_run_program(edmp_ctx* ctx, int B, int t, const TailP* tail, bool* tail_done) {
    if (tail_done) *tail_done = false;
    UNet* u = ctx->unet;
    EDMP_REQUIRE(u, "edmp_unet_load has not been called");
    EDMP_REQUIRE(B >= 1 && B <= u->max_batch, "batch %d outside 1..max_batch=%d", B, u->max_batch);
    hipStream_t main_stream = ctx->stream;
    EDMP_REQUIRE(t >= 1 && t <= u->desc.T, "t=%d outside 1..T=%d", t, u->desc.T);
    const float* trow = u->tbias + (size_t)(t - 1) * u->tb_stride;
    Prof& pf = ctx->prof;
    Prof::Pend whole{nullptr, nullptr, -1};
    if (pf.on == 2) {  // one bracket around the whole program: the conv family's time with no per-launch event overhead
        if (!pf.pool.empty()) {
            whole.a = pf.pool.back().first;
            whole.b = pf.pool.back().second;
            pf.pool.pop_back();
        } else {
            EDMP_HIP_CHECK(hipEventCreate(&whole.a));
            EDMP_HIP_CHECK(hipEventCreate(&whole.b));
        }
        EDMP_HIP_CHECK(hipEventRecord(whole.a, main_stream));
    }
    // the tail of the reverse step runs inside the last level's launch when that level is the program's last op (LV_UP_FINAL, 32 channels)
    auto fuse_step_tail = [&](LevelP& p, const Op& o, int index, bool out_is_head_input) {
        if (tail && tail_done && o.lv_variant == 4 && index + 1 == (int)u->prog.size() && u->head_cin == 32 && tail->N == u->desc.horizon && tail->C <= 8 && out_is_head_input) {
            p.tail = *tail;
            p.tail.on = 1;
            p.tail.w = u->head_w;
            p.tail.bias = u->head_b;
            *tail_done = true;
        }
    };
    int op_index = -1, skip_index = -1;
    for (const Op& op : u->prog) {
        ++op_index;
        if (op_index == skip_index) continue;  // the second level of a merged pair: ran inside the previous launch
        hipStream_t s = main_stream;
        Prof::Pend ev{nullptr, nullptr, op_index};
        const bool timed = pf.on == 1 && op.kind != OP_GN;
        if (timed) {
            if (!pf.pool.empty()) {
                ev.a = pf.pool.back().first;
                ev.b = pf.pool.back().second;
                pf.pool.pop_back();
            } else {
                EDMP_HIP_CHECK(hipEventCreate(&ev.a));
                EDMP_HIP_CHECK(hipEventCreate(&ev.b));
            }
            EDMP_HIP_CHECK(hipEventRecord(ev.a, s));
        }
        int rc = EDMP_OK;
        if (op.kind == OP_RCB) {
            RcbP p = op.rc;
            p.B = B;
            p.add_tb = op.tb_off >= 0 ? trow + op.tb_off : nullptr;
            EDMP_REQUIRE(!(p.add_tb && p.add_res), "fused conv block: a launch adds the time bias (conv1) or the residual (conv2), not both");
            rc = launch_rcb(p, op.rc_L, op.rc_form, op.rc_ms, op.rc_bf3, s);
        } else if (op.kind == OP_LVL && op.lv_merge) {
            const Op& nx = u->prog[op_index + 1];
            LevelP pa = op.lv, pb = nx.lv;
            pa.B = pb.B = B;
            const bool out_is_head_input = pb.out == u->h_last;
            pa.out = nullptr, pb.src1 = nullptr;  // (level B's first input half arrives in LDS)
            pa.tb1 = trow + op.lv_tb1, pa.tb2 = trow + op.lv_tb2;
            pb.tb1 = trow + nx.lv_tb1, pb.tb2 = trow + nx.lv_tb2;
            fuse_step_tail(pb, nx, op_index + 1, out_is_head_input);
            rc = launch_level2(pa, pb, op.lv_variant, nx.lv_variant, s);
            skip_index = op_index + 1;
        } else if (op.kind == OP_LVL) {
            LevelP p = op.lv;
            p.B = B;
            const bool out_is_head_input = p.out == u->h_last;
            p.tb1 = trow + op.lv_tb1;
            p.tb2 = trow + op.lv_tb2;
            fuse_step_tail(p, op, op_index, out_is_head_input);
            rc = launch_level(p, op.lv_variant, op.lv_sb, s);
        } else if (op.kind == OP_WRS) {
            RcbP p = op.rc;
            p.B = B;
            rc = launch_wrs(p, op.wrs_kind, op.rc_L, op.rc_ms, op.rc_bf3, s);
        } else if (op.kind == OP_CONV) {
            ConvP p = op.cv;
            p.B = B;
            launch_conv(p, s);
        } else {
            GnP g = op.gn;
            g.B = B;
            g.add_tbias = op.tb_off >= 0 ? trow + op.tb_off : nullptr;
            rc = launch_gn(g, s);
        }
        if (rc) return rc;
        if (timed) {
            EDMP_HIP_CHECK(hipEventRecord(ev.b, s));
            pf.pending.push_back(ev);
        }
    }
    if (pf.on == 2) {
        EDMP_HIP_CHECK(hipEventRecord(whole.b, main_stream));
        pf.pending.push_back(whole);
    }
    EDMP_HIP_CHECK(hipGetLastError());
    return EDMP_OK;
}

int prof_fold(edmp_ctx* ctx) {
    Prof& p = ctx->prof;
    const size_t nops = ctx->unet ? ctx->unet->prog.size() : 0;
    if (p.op_ms.size() != nops) {
        p.op_ms.assign(nops, 0.0);
        p.op_calls.assign(nops, 0);
    }
    for (auto& e : p.pending) {
        float ms = 0.f;
        EDMP_HIP_CHECK(hipEventElapsedTime(&ms, e.a, e.b));
        p.conv_ms += ms;
        if (e.op < 0) {  // whole-program bracket: counts every launch of the program
            if (ctx->unet)
            {
                bool second_of_pair = false;  // (the second level of a merged pair is not a launch of its own)
                for (const Op& op : ctx->unet->prog) {
                    p.conv_launches += (op.kind != OP_GN && !second_of_pair) ? 1 : 0;
                    second_of_pair = op.kind == OP_LVL && op.lv_merge;
                }
            }
        } else {
            p.conv_launches += 1;
        }
        if (e.op >= 0 && (size_t)e.op < nops) {
            p.op_ms[e.op] += ms;
            p.op_calls[e.op] += 1;
        }
        p.pool.push_back({e.a, e.b});
    }
    p.pending.clear();
    return EDMP_OK;
}

int unet_forward_impl(edmp_ctx* ctx, const float* x_dev, int B, int t, float* eps_dev) {
    UNet* u = ctx->unet;
    EDMP_REQUIRE(u, "edmp_unet_load has not been called");
    EDMP_REQUIRE(B >= 1 && B <= u->max_batch, "batch %d outside 1..max_batch=%d", B, u->max_batch);
    hipStream_t s = ctx->stream;
    const int N = u->desc.horizon, C = u->desc.input_dim;
    {
        int total = B * N * 8;
        hipLaunchKernelGGL(pack_input_kernel, dim3((total + 255) / 256), dim3(256), 0, s, x_dev, u->x_in, B, C, N, 8);
    }
    int rc = unet_run_program(ctx, B, t, nullptr, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(head_1x1_kernel, dim3((B * N + 255) / 256), dim3(256), 0, s, u->h_last, u->head_w, u->head_b, eps_dev, B, N,
                       u->head_cin, C);
    EDMP_HIP_CHECK(hipGetLastError());
    return EDMP_OK;
}
}  // namespace edmp

extern "C" int edmp_unet_forward_dev(edmp_ctx* ctx, const float* x_dev, int B, int t, float* eps_dev) {
    EDMP_REQUIRE(ctx && x_dev && eps_dev, "edmp_unet_forward_dev: null argument");
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    return unet_forward_impl(ctx, x_dev, B, t, eps_dev);
}

extern "C" int edmp_unet_read_activation_dev(edmp_ctx* ctx, int which, int B, float* out_dev, int* C_out, int* L_out) {
    EDMP_REQUIRE(ctx && ctx->unet && out_dev, "edmp_unet_read_activation_dev: null argument / no model");
    for (auto& t : ctx->unet->taps)
        if (t.which == which) {
            int total = B * t.C * t.L;
            hipLaunchKernelGGL(unpack_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, t.p, out_dev, B, t.L, t.C);
            EDMP_HIP_CHECK(hipGetLastError());
            if (C_out) *C_out = t.C;
            if (L_out) *L_out = t.L;
            return EDMP_OK;
        }
    set_error("no activation tap %d", which);
    return EDMP_ERR_ARG;
}

extern "C" int edmp_unet_flops_direct(edmp_ctx* ctx, double* direct) {
    EDMP_REQUIRE(ctx && ctx->unet && direct, "no model loaded");
    *direct = ctx->unet->flops_direct;
    return EDMP_OK;
}

extern "C" int edmp_unet_flops(edmp_ctx* ctx, double* nominal, double* executed) {
    EDMP_REQUIRE(ctx && ctx->unet, "no model loaded");
    if (nominal) *nominal = ctx->unet->flops_nominal;
    if (executed) *executed = ctx->unet->flops_exec;
    return EDMP_OK;
}

extern "C" int edmp_unet_flops_pipes(edmp_ctx* ctx, double* f32_issued, double* bf16_issued) {
    EDMP_REQUIRE(ctx && ctx->unet, "no model loaded");
    if (f32_issued) *f32_issued = ctx->unet->flops_exec - ctx->unet->flops_f32_moved;
    if (bf16_issued) *bf16_issued = ctx->unet->flops_bf16;
    return EDMP_OK;
}

extern "C" int edmp_prof_ops_bf16(edmp_ctx* ctx, int cap, int* n_ops, double* flops_bf16) {
    EDMP_REQUIRE(ctx && ctx->unet && n_ops, "edmp_prof_ops_bf16: null argument / no model");
    const int n = (int)ctx->unet->prog.size();
    *n_ops = n;
    for (int i = 0; i < n && i < cap; ++i)
        if (flops_bf16) flops_bf16[i] = ctx->unet->prog[i].flops_bf16;
    return EDMP_OK;
}

extern "C" int edmp_prof_ops(edmp_ctx* ctx, int cap, int* n_ops, double* ms, int64_t* calls, double* flops_exec, char* names) {
    EDMP_REQUIRE(ctx && ctx->unet && n_ops, "edmp_prof_ops: null argument / no model");
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    EDMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (int rc = prof_fold(ctx)) return rc;
    const int n = (int)ctx->unet->prog.size();
    *n_ops = n;
    for (int i = 0; i < n && i < cap; ++i) {
        const Op& op = ctx->unet->prog[i];
        if (ms) ms[i] = ctx->prof.op_ms[i];
        if (calls) calls[i] = ctx->prof.op_calls[i];
        if (flops_exec) flops_exec[i] = op.flops_exec;
        if (names) {
            strncpy(names + (size_t)i * 64, op.name, 63);
            names[(size_t)i * 64 + 63] = 0;
        }
    }
    return EDMP_OK;
}

#ifdef EDMP_STAMPS
extern "C" int edmp_debug_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(edmp::g_stamps), sizeof(unsigned long long) * 8 * 16);
}
#endif
