// Round-1 wide fused conv kernel (weights staged through LDS), kept ONLY as the A/B reference of tools/kbench.hip:
// the library uses wide_conv_kernel (edmp_amd/csrc/wide.hip).  Include after libedmp_hip.hip.
namespace edmp {
// ---------------------------------------------------------------------------------------------------------------
// Fused Conv1dBlock of the wide levels: Conv1d(k=5, pad=2) + bias -> GroupNorm(8) -> Mish -> (+ time-bias | + residual)
// (blocks.py:22-28 and the adds of blocks.py:162-164) in ONE launch, for Cout/8 in {32, 64} and L in {2, 4, 7}.
//
// A workgroup owns 32 samples x ONE GroupNorm group (CG channels) x ALL L output positions, so the normalisation
// statistics are complete inside the workgroup and the raw convolution output never travels to HBM.
// K step = 32 input channels of ALL L input positions (A: [L][32][36] floats) + the CG x 32 weight slab of every tap
// that can be valid (B: [taps][CG][36]); a wave owns up to two 32x32 output tiles (position l, 32-channel slab) and,
// per step, runs 16 MFMAs for every input position within +-2 of its l (taps in the zero padding do not exist).
// That is 32..112 MFMAs per wave per barrier instead of 16, and 8x less activation re-reading than the per-position
// tiling of conv_mfma_kernel.  grid = (8 groups, B/32): blockIdx.x = group, so one XCD's L2 holds one group's weights.


template <int CG, int L, bool RES = false>
struct RcbCfg {
    static constexpr int KC = 32, LDK = KC + 4;
    static constexpr int S = CG / 32;
    static constexpr int NTILE = L * S;
    static constexpr int NT = (NTILE + 3) / 4;
    static constexpr int KT0 = (L == 2) ? 1 : 0;
    static constexpr int NTAP = (L == 2) ? 3 : 5;
    static constexpr int NSLAB = NTAP + (RES ? 1 : 0);  // weight slabs per K step: the valid taps (+ the residual 1x1 conv)
    static constexpr int A_FL = L * 32 * LDK;
    static constexpr int B_FL = NSLAB * CG * LDK;
    static constexpr int STAGE = A_FL + B_FL;
    static constexpr int NA = L;
    static constexpr int NB = NSLAB * CG / 32;
    static constexpr int YS = L * CG + 4;
    static constexpr int NF4 = L * CG / 32;  // float4 per thread in the epilogue
    // Split-K mode: instead of owning output tiles, a wave owns a SLICE of each 32-channel K chunk (8 channels with one
    // 32-channel output slab per group, 16 with two) and accumulates ALL L tiles of its slab over it.  All waves then
    // run the same, fully static instruction stream (no per-wave tile/position loops, no imbalance between edge and
    // centre positions), the A fragment of an input position is loaded once and reused by every tile it feeds, and
    // consecutive MFMA groups go to different accumulators.  The partial tiles are summed when the epilogue reads them
    // back from LDS.
#ifdef EDMP_NO_SPLITK
    static constexpr bool SK = false;
#else
    // used where a group is one 32-channel slab: there the per-position tile ownership leaves edge-position waves idle
    // (3 vs 4 or 5 taps).  Measured: <32,7> 44.8 -> 40.4 us, <32,4> 23.8 -> 22.1 us, <32,4,RES> 97.7 -> 91 us.  With two
    // slabs the tile ownership is already balanced: split-K measured slower at L = 2 (24.2 -> 25.8 us) and spills at L = 4;
    // the L = 7 residual variant (14 accumulators) spills too.
    static constexpr bool SK = (S == 1) && !(RES && L > 4);
#endif
    // a wave = (output slab s = wave % S, K slice ks = wave / S): KSPLIT waves share a slab, each taking QW of the
    // chunk's four 8-channel K groups
    static constexpr int KSPLIT = SK ? 4 / S : 1;
    static constexpr int QW = 4 / (4 / S);
    static constexpr int NP = SK ? KSPLIT : 1;   // partial output tiles in LDS
    static constexpr int NACC = SK ? L : NT;     // accumulators per wave
    // the MFMA groups (4 MFMAs each) of one split-K step: for every input position lp and every K group q of the wave,
    // the tiles l with |l - lp| <= 2 (weight slab lp - l + 2 - KT0), then the residual conv of tile lp (slab NTAP) -
    // so the A fragment (lp, q) is loaded once and feeds up to six groups
    static constexpr int sk_groups() {
        int n = 0;
        for (int lp = 0; lp < L; ++lp)
            for (int q = 0; q < QW; ++q) {
                for (int l = (lp - 2 > 0 ? lp - 2 : 0); l <= (lp + 2 < L - 1 ? lp + 2 : L - 1); ++l) ++n;
                if (RES) ++n;
            }
        return n;
    }
    // what = 0: input position, 1: output tile, 2: weight slab, 3: is-residual, 4: K group q
    static constexpr int sk_group(int g, int what) {
        int n = 0;
        for (int lp = 0; lp < L; ++lp)
            for (int q = 0; q < QW; ++q) {
                for (int l = (lp - 2 > 0 ? lp - 2 : 0); l <= (lp + 2 < L - 1 ? lp + 2 : L - 1); ++l) {
                    if (n == g) return what == 0 ? lp : what == 1 ? l : what == 2 ? lp - l + 2 - KT0 : what == 3 ? 0 : q;
                    ++n;
                }
                if (RES) {
                    if (n == g) return what == 0 ? lp : what == 1 ? lp : what == 2 ? NTAP : what == 3 ? 1 : q;
                    ++n;
                }
            }
        return 0;
    }
    static constexpr size_t lds_bytes() {
        size_t b = 2 * (size_t)STAGE * sizeof(float);
        size_t y = (size_t)NP * 32 * (size_t)YS * sizeof(float);
        size_t m = b > y ? b : y;
#ifdef EDMP_EXP_NOPIN
        return m;
#else
        return m > 83968 ? m : 83968;  // > 80 KiB: at most one workgroup per CU, so 256 workgroups cover 256 CUs
#endif
    }
};

template <int CG, int L, bool RES>
__global__ __launch_bounds__(256) void rcb_conv_kernel(RcbP p) {
    using Cf = RcbCfg<CG, L, RES>;
    constexpr int KC = Cf::KC, LDK = Cf::LDK, S = Cf::S, NTILE = Cf::NTILE, NT = Cf::NT, KT0 = Cf::KT0, NTAP = Cf::NTAP;
    constexpr bool SPK = Cf::SK;  // split-K mode
    constexpr int NP = Cf::NP, NACC = Cf::NACC;
    constexpr int A_FL = Cf::A_FL, STAGE = Cf::STAGE, NA = Cf::NA, NB = Cf::NB, YS = Cf::YS, NF4 = Cf::NF4;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    constexpr int SK = (L == 2) ? 2 : (L == 4 && CG == 64) ? 3 : (L == 7) ? 4 : 5;
    EDMP_STAMP(SK, 0)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int co0 = blockIdx.x * CG;
    const int b0 = blockIdx.y * 32;
    const int Cin = p.C1 + p.C2;
    const int ch1 = p.C1 / KC, ch2 = p.C2 / KC;
    const int nK = ch1 + ch2;

    // staging maps (chunk invariant)
    const int srow = tid >> 3, sc4 = (tid & 7) * 4;
    const int sb = min(b0 + srow, p.B - 1);
    const int a_g1 = sb * L * p.C1 + sc4;  // + l'*C1 + ci0
    const int a_g2 = sb * L * p.C2 + sc4;
    const int a_l = srow * LDK + sc4;      // + l'*(32*LDK)
    // staging registers are individual scalars: hipcc parks small arrays in scratch memory once scheduling
    // barriers pin the prefetch (seen with ROCm 7.2), which would serialise every load behind a vmcnt(0)
#define EDMP_REP7(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6)
#define EDMP_REP10(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11)
    static_assert(NA <= 7 && NB <= 12, "staging macros cover NA <= 7, NB <= 12");
    // weight slab of staging item i: slabs 0..NTAP-1 are the conv taps KT0.., slab NTAP (RES) is the residual 1x1 conv,
    // stored as tap index 5 of the packed weight tensor
#define EDMP_DECL_RB(i)                                                                              \
    const int wg##i = ((((tid + i * 256) / (CG * 8)) < NTAP ? KT0 + (tid + i * 256) / (CG * 8) : 5) * p.Cout + co0 + (((tid + i * 256) % (CG * 8)) >> 3)) * Cin + sc4; \
    const int bl##i = A_FL + (((tid + i * 256) / (CG * 8)) * CG + (((tid + i * 256) % (CG * 8)) >> 3)) * LDK + sc4;          \
    float4 rbP##i = make_float4(0.f, 0.f, 0.f, 0.f), rbQ##i = make_float4(0.f, 0.f, 0.f, 0.f);
#define EDMP_DECL_RA(i) float4 raP##i = make_float4(0.f, 0.f, 0.f, 0.f), raQ##i = make_float4(0.f, 0.f, 0.f, 0.f);
    EDMP_REP7(EDMP_DECL_RA)
    EDMP_REP10(EDMP_DECL_RB)
#define EDMP_LD_AP(i) \
    if constexpr (i < NA) raP##i = *reinterpret_cast<const float4*>(src_ + ag_ + i * Cs_);
#define EDMP_LD_BP(i) \
    if constexpr (i < NB) rbP##i = *reinterpret_cast<const float4*>(p.W + wg##i + wofs_);
#define EDMP_ST_AP(i) \
    if constexpr (i < NA) *reinterpret_cast<float4*>(sn_ + i * (32 * LDK) + a_l) = raP##i;
#define EDMP_ST_BP(i) \
    if constexpr (i < NB) *reinterpret_cast<float4*>(sn_ + bl##i) = rbP##i;
#define EDMP_LD_AQ(i) \
    if constexpr (i < NA) raQ##i = *reinterpret_cast<const float4*>(src_ + ag_ + i * Cs_);
#define EDMP_LD_BQ(i) \
    if constexpr (i < NB) rbQ##i = *reinterpret_cast<const float4*>(p.W + wg##i + wofs_);
#define EDMP_ST_AQ(i) \
    if constexpr (i < NA) *reinterpret_cast<float4*>(sn_ + i * (32 * LDK) + a_l) = raQ##i;
#define EDMP_ST_BQ(i) \
    if constexpr (i < NB) *reinterpret_cast<float4*>(sn_ + bl##i) = rbQ##i;
// fetch chunk `nc` (of the ch1 + ch2 channel chunks) into the staging registers
// fetch channel chunk `nc` into register set P or Q (the K loop keeps TWO chunks in flight: first-touch activation
// rows come from MALL/HBM at ~1 us, longer than one K step)
#define EDMP_RCB_FETCH(SET, nc)                                        \
    {                                                                  \
        const bool first_ = (nc) < ch1;                                \
        const float* src_ = first_ ? p.src1 : p.src2;                  \
        const int Cs_ = first_ ? p.C1 : p.C2;                          \
        const int ci0_ = (first_ ? (nc) : (nc)-ch1) * KC;              \
        const int ag_ = (first_ ? a_g1 : a_g2) + ci0_;                 \
        const int wofs_ = (first_ ? 0 : p.C1) + ci0_;                  \
        EDMP_REP7(EDMP_LD_A##SET) EDMP_REP10(EDMP_LD_B##SET)           \
    }
#define EDMP_RCB_COMMIT(SET, stage_ptr)                        \
    {                                                          \
        float* sn_ = (stage_ptr);                              \
        EDMP_REP7(EDMP_ST_A##SET) EDMP_REP10(EDMP_ST_B##SET)   \
    }
    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    // the conv bias of this wave's tiles is requested NOW: loaded where it is used (first thing of the epilogue) it would
    // sit behind the epilogue's own operand prefetches in the in-order vmcnt queue and stall the accumulator spill ~1 us
    static_assert(NT <= 2, "bias registers cover two tiles per wave");
    const float bias_t0 = p.bias[co0 + (wave % S) * 32 + (lane & 31)];
    const float bias_t1 = p.bias[co0 + (min(wave + 4, NTILE - 1) % S) * 32 + (lane & 31)];
    // folded residual 1x1 conv (RES): a second accumulator per tile, fed by the centre input position against weight slab NTAP
    f32x16 racc[RES ? NACC : 1];
    float rbias_t0 = 0.f, rbias_t1 = 0.f;
    if constexpr (RES) {
#pragma unroll
        for (int t = 0; t < NACC; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) racc[t][i] = 0.0f;
        rbias_t0 = p.res_bias[co0 + (wave % S) * 32 + (lane & 31)];
        rbias_t1 = p.res_bias[co0 + (min(wave + 4, NTILE - 1) % S) * 32 + (lane & 31)];
    }

    // prologue: chunk 0 -> stage 0, chunk 1 in flight in set Q
    EDMP_RCB_FETCH(P, 0)
    if (nK > 1) EDMP_RCB_FETCH(Q, 1)
    EDMP_RCB_COMMIT(P, lds)
    __syncthreads();
    EDMP_STAMP(SK, 1)

    const int frag = (lane & 31) * LDK + 4 * (lane >> 5);

// the residual 1x1 conv of tile (l, s): centre position l of the A stage x weight slab NTAP, 16 MFMAs into racc[t]
#define EDMP_RCB_RESID(st, t, l, s)                                                                            \
    if constexpr (RES) {                                                                                       \
        const float* ar_ = (st) + (l) * (32 * LDK) + frag;                                                     \
        const float* br_ = (st) + A_FL + (NTAP * CG + (s) * 32) * LDK + frag;                                  \
        _Pragma("unroll") for (int q = 0; q < KC / 8; ++q) {                                                   \
            const float4 ra4 = *reinterpret_cast<const float4*>(ar_ + 8 * q);                                 \
            const float4 rb4 = *reinterpret_cast<const float4*>(br_ + 8 * q);                                 \
            racc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra4.x, rb4.x, racc[t], 0, 0, 0);                    \
            racc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra4.y, rb4.y, racc[t], 0, 0, 0);                    \
            racc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra4.z, rb4.z, racc[t], 0, 0, 0);                    \
            racc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra4.w, rb4.w, racc[t], 0, 0, 0);                    \
        }                                                                                                      \
    }
// all MFMAs of one K step for this wave's tiles, reading stage `st`.  The A/B fragments of the NEXT group of four
// MFMAs are requested from LDS before the current four are issued (software pipelining by hand: one wave per SIMD
// has nobody else to hide the ds_read latency).
#define EDMP_RCB_COMPUTE(st)                                                                                   \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                           \
        const int j = wave + 4 * t;                                                                            \
        if (j < NTILE) {                                                                                       \
            const int l = j / S, s = j % S;                                                                    \
            const int lp_lo = max(0, l - 2), lp_hi = min(L - 1, l + 2);                                        \
            const float* a_s = (st) + lp_lo * (32 * LDK) + frag;                                               \
            const float* b_s = (st) + A_FL + ((lp_lo - l + 2 - KT0) * CG + s * 32) * LDK + frag;               \
            float4 a4 = *reinterpret_cast<const float4*>(a_s);                                                 \
            float4 b4 = *reinterpret_cast<const float4*>(b_s);                                                 \
            for (int lp = lp_lo; lp <= lp_hi; ++lp) {                                                          \
                const int adv = (lp < lp_hi) ? 1 : 0;                                                          \
                _Pragma("unroll") for (int q = 0; q < KC / 8; ++q) {                                           \
                    const float* an = (q < KC / 8 - 1) ? a_s + 8 * (q + 1) : a_s + adv * (32 * LDK);           \
                    const float* bn = (q < KC / 8 - 1) ? b_s + 8 * (q + 1) : b_s + adv * (CG * LDK);           \
                    const float4 a4n = *reinterpret_cast<const float4*>(an);                                   \
                    const float4 b4n = *reinterpret_cast<const float4*>(bn);                                   \
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[t], 0, 0, 0);                \
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[t], 0, 0, 0);                \
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[t], 0, 0, 0);                \
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[t], 0, 0, 0);                \
                    a4 = a4n;                                                                                  \
                    b4 = b4n;                                                                                  \
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); /* 2 ds_read (next fragments) ... */    \
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); /* ... ahead of the 4 current MFMAs */  \
                }                                                                                              \
                a_s += 32 * LDK;                                                                               \
                b_s += CG * LDK;                                                                               \
            }                                                                                                  \
            EDMP_RCB_RESID(st, t, l, s)                                                                        \
        }                                                                                                      \
    }

    // ---- steady state ----------------------------------------------------------------------------------------------
    // One K step = MFMAs on the current stage + global fetch of the chunk after next + LDS commit of the next chunk.
    // The three are independent, but one wave per SIMD only overlaps what its own instruction stream interleaves: a
    // phase-timed build showed fetch + commit + barrier serialised after the MFMAs cost 13-32 % of a step.  So the
    // first 16 MFMAs of every step (tile 0, first input position - present for every wave) carry the step's memory
    // instructions in their issue gaps: after each MFMA one or two global loads and one or two ds_writes.
    // Unrolled by two so that the register sets alternate statically:
    //   even step: MFMAs on stage 0 (chunk kk)   | set P fetches chunk kk+2 | set Q (chunk kk+1) -> stage 1
    //   odd  step: MFMAs on stage 1 (chunk kk+1) | set Q fetches chunk kk+3 | set P (chunk kk+2) -> stage 0
    // Fetches past the last chunk re-read the last chunk and commits of them land in a stage nobody reads again
    // (unconditional straight-line code is what lets the scheduler interleave).
#define EDMP_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0);
#define EDMP_RCB_MFMA4(t)                                                                  \
    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[t], 0, 0, 0);           \
    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[t], 0, 0, 0);           \
    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[t], 0, 0, 0);           \
    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[t], 0, 0, 0);
// memory work of one 4-MFMA group q (0..3): items i with i % 4 == q of the A (<= 7) and B (<= 12) staging lists
#define EDMP_RCB_MEM0(LS, SS) EDMP_LD_A##LS(0) EDMP_LD_A##LS(4) EDMP_LD_B##LS(0) EDMP_LD_B##LS(4) EDMP_LD_B##LS(8) \
                              EDMP_ST_A##SS(0) EDMP_ST_A##SS(4) EDMP_ST_B##SS(0) EDMP_ST_B##SS(4) EDMP_ST_B##SS(8)
#define EDMP_RCB_MEM1(LS, SS) EDMP_LD_A##LS(1) EDMP_LD_A##LS(5) EDMP_LD_B##LS(1) EDMP_LD_B##LS(5) EDMP_LD_B##LS(9) \
                              EDMP_ST_A##SS(1) EDMP_ST_A##SS(5) EDMP_ST_B##SS(1) EDMP_ST_B##SS(5) EDMP_ST_B##SS(9)
#define EDMP_RCB_MEM2(LS, SS) EDMP_LD_A##LS(2) EDMP_LD_A##LS(6) EDMP_LD_B##LS(2) EDMP_LD_B##LS(6) EDMP_LD_B##LS(10) \
                              EDMP_ST_A##SS(2) EDMP_ST_A##SS(6) EDMP_ST_B##SS(2) EDMP_ST_B##SS(6) EDMP_ST_B##SS(10)
#define EDMP_RCB_MEM3(LS, SS) EDMP_LD_A##LS(3) EDMP_LD_B##LS(3) EDMP_LD_B##LS(7) EDMP_LD_B##LS(11) \
                              EDMP_ST_A##SS(3) EDMP_ST_B##SS(3) EDMP_ST_B##SS(7) EDMP_ST_B##SS(11)
#define EDMP_RCB_QMEM(q, LS, SS)                                                                        \
    {                                                                                                   \
        const float* an = (q < KC / 8 - 1) ? a_s + 8 * (q + 1) : a_s + (32 * LDK);                      \
        const float* bn = (q < KC / 8 - 1) ? b_s + 8 * (q + 1) : b_s + (CG * LDK);                      \
        const float4 a4n = *reinterpret_cast<const float4*>(an);                                        \
        const float4 b4n = *reinterpret_cast<const float4*>(bn);                                        \
        EDMP_RCB_MEM##q(LS, SS)                                                                         \
        EDMP_RCB_MFMA4(0)                                                                               \
        a4 = a4n;                                                                                       \
        b4 = b4n;                                                                                       \
        EDMP_SGB(0x100, 2)                                          /* next fragments */                \
        EDMP_SGB(0x008, 1) EDMP_SGB(0x020, 2) EDMP_SGB(0x200, 2)    /* MFMA | 2 loads | 2 ds_writes */  \
        EDMP_SGB(0x008, 1) EDMP_SGB(0x020, 1) EDMP_SGB(0x200, 1)                                        \
        EDMP_SGB(0x008, 1) EDMP_SGB(0x020, 1) EDMP_SGB(0x200, 1)                                        \
        EDMP_SGB(0x008, 1) EDMP_SGB(0x020, 1) EDMP_SGB(0x200, 1)                                        \
    }
// the remaining input positions of a tile (lp_from .. lp_hi), fragments of the next group requested one group ahead
#define EDMP_RCB_TAPS(t, lp_from)                                                                              \
    for (int lp = (lp_from); lp <= lp_hi; ++lp) {                                                              \
        const int adv = (lp < lp_hi) ? 1 : 0;                                                                  \
        _Pragma("unroll") for (int q = 0; q < KC / 8; ++q) {                                                   \
            const float* an = (q < KC / 8 - 1) ? a_s + 8 * (q + 1) : a_s + adv * (32 * LDK);                   \
            const float* bn = (q < KC / 8 - 1) ? b_s + 8 * (q + 1) : b_s + adv * (CG * LDK);                   \
            const float4 a4n = *reinterpret_cast<const float4*>(an);                                           \
            const float4 b4n = *reinterpret_cast<const float4*>(bn);                                           \
            EDMP_RCB_MFMA4(t)                                                                                  \
            a4 = a4n;                                                                                          \
            b4 = b4n;                                                                                          \
            EDMP_SGB(0x100, 2) /* 2 ds_read (next fragments) ... */                                            \
            EDMP_SGB(0x008, 4) /* ... ahead of the 4 current MFMAs */                                          \
        }                                                                                                      \
        a_s += 32 * LDK;                                                                                       \
        b_s += CG * LDK;                                                                                       \
    }
#define EDMP_RCB_TILE_SETUP(st, j)                                                                             \
    const int l = (j) / S, s = (j) % S;                                                                        \
    const int lp_lo = max(0, l - 2), lp_hi = min(L - 1, l + 2);                                                \
    const float* a_s = (st) + lp_lo * (32 * LDK) + frag;                                                       \
    const float* b_s = (st) + A_FL + ((lp_lo - l + 2 - KT0) * CG + s * 32) * LDK + frag;                       \
    float4 a4 = *reinterpret_cast<const float4*>(a_s);                                                         \
    float4 b4 = *reinterpret_cast<const float4*>(b_s);
// one K step: MFMAs on stage `st`; register set LS fetches chunk `nc`; register set SS is committed to stage `stn`
#define EDMP_RCB_STEP(st, LS, nc, SS, stn)                                                                     \
    {                                                                                                          \
        const int nc_ = min((nc), nK - 1);                                                                     \
        const bool first_ = nc_ < ch1;                                                                         \
        const float* src_ = first_ ? p.src1 : p.src2;                                                          \
        const int Cs_ = first_ ? p.C1 : p.C2;                                                                  \
        const int ci0_ = (first_ ? nc_ : nc_ - ch1) * KC;                                                      \
        const int ag_ = (first_ ? a_g1 : a_g2) + ci0_;                                                         \
        const int wofs_ = (first_ ? 0 : p.C1) + ci0_;                                                          \
        float* sn_ = (stn);                                                                                    \
        {                                                                                                      \
            EDMP_RCB_TILE_SETUP(st, wave)                                                                      \
            EDMP_RCB_QMEM(0, LS, SS) EDMP_RCB_QMEM(1, LS, SS) EDMP_RCB_QMEM(2, LS, SS) EDMP_RCB_QMEM(3, LS, SS) \
            a_s += 32 * LDK;                                                                                   \
            b_s += CG * LDK;                                                                                   \
            EDMP_RCB_TAPS(0, lp_lo + 1)                                                                        \
            EDMP_RCB_RESID(st, 0, l, s)                                                                        \
        }                                                                                                      \
        _Pragma("unroll") for (int t = 1; t < NT; ++t) {                                                       \
            const int j = wave + 4 * t;                                                                        \
            if (j < NTILE) {                                                                                   \
                EDMP_RCB_TILE_SETUP(st, j)                                                                     \
                EDMP_RCB_TAPS(t, lp_lo)                                                                        \
                EDMP_RCB_RESID(st, t, l, s)                                                                    \
            }                                                                                                  \
        }                                                                                                      \
    }
    static_assert(KC == 32 && NTILE >= 4 && L >= 2, "EDMP_RCB_STEP: 4 MFMA groups per position, a tile 0 with >= 2 positions per wave");
// ---- split-K step (Cf::SK): this wave's K slice = channels [8*wave, 8*wave+8) of the chunk; the groups of Cf::sk_group
// in order, fragments of group g+1 requested before the MFMAs of group g, the step's staging traffic on the first four
#define EDMP_SK_MEMSEL(g, LS, SS)                      \
    if ((g) == 0) { EDMP_RCB_MEM0(LS, SS) }            \
    else if ((g) == 1) { EDMP_RCB_MEM1(LS, SS) }       \
    else if ((g) == 2) { EDMP_RCB_MEM2(LS, SS) }       \
    else if ((g) == 3) { EDMP_RCB_MEM3(LS, SS) }
#define EDMP_SK_STEP(st, LS, nc, SS, stn)                                                                      \
    {                                                                                                          \
        const int nc_ = min((nc), nK - 1);                                                                     \
        const bool first_ = nc_ < ch1;                                                                         \
        const float* src_ = first_ ? p.src1 : p.src2;                                                          \
        const int Cs_ = first_ ? p.C1 : p.C2;                                                                  \
        const int ci0_ = (first_ ? nc_ : nc_ - ch1) * KC;                                                      \
        const int ag_ = (first_ ? a_g1 : a_g2) + ci0_;                                                         \
        const int wofs_ = (first_ ? 0 : p.C1) + ci0_;                                                          \
        float* sn_ = (stn);                                                                                    \
        const float* sa_ = (st) + frag + 8 * Cf::QW * (wave / S);                                              \
        const float* sb_ = (st) + A_FL + ((wave % S) * 32) * LDK + frag + 8 * Cf::QW * (wave / S);             \
        float4 a4 = *reinterpret_cast<const float4*>(sa_ + Cf::sk_group(0, 0) * (32 * LDK) + 8 * Cf::sk_group(0, 4)); \
        float4 b4 = *reinterpret_cast<const float4*>(sb_ + Cf::sk_group(0, 2) * (CG * LDK) + 8 * Cf::sk_group(0, 4)); \
        _Pragma("unroll") for (int g = 0; g < NGRP; ++g) {                                                     \
            const int tl_ = Cf::sk_group(g, 1);                                                                \
            const bool isres_ = Cf::sk_group(g, 3) != 0;                                                       \
            float4 a4n = a4, b4n = b4;                                                                         \
            if (g + 1 < NGRP) {                                                                                \
                if (Cf::sk_group(g + 1, 0) != Cf::sk_group(g, 0) || Cf::sk_group(g + 1, 4) != Cf::sk_group(g, 4)) \
                    a4n = *reinterpret_cast<const float4*>(sa_ + Cf::sk_group(g + 1, 0) * (32 * LDK) + 8 * Cf::sk_group(g + 1, 4)); \
                b4n = *reinterpret_cast<const float4*>(sb_ + Cf::sk_group(g + 1, 2) * (CG * LDK) + 8 * Cf::sk_group(g + 1, 4));     \
            }                                                                                                  \
            EDMP_SK_MEMSEL(g, LS, SS)                                                                          \
            if (isres_) {                                                                                      \
                if constexpr (RES) {                                                                           \
                    racc[tl_] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, racc[tl_], 0, 0, 0);          \
                    racc[tl_] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, racc[tl_], 0, 0, 0);          \
                    racc[tl_] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, racc[tl_], 0, 0, 0);          \
                    racc[tl_] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, racc[tl_], 0, 0, 0);          \
                }                                                                                              \
            } else {                                                                                           \
                acc[tl_] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[tl_], 0, 0, 0);                \
                acc[tl_] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[tl_], 0, 0, 0);                \
                acc[tl_] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[tl_], 0, 0, 0);                \
                acc[tl_] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[tl_], 0, 0, 0);                \
            }                                                                                                  \
            a4 = a4n;                                                                                          \
            b4 = b4n;                                                                                          \
            if (g < 4) {                                                                                       \
                EDMP_SGB(0x100, 2)                                                                             \
                EDMP_SGB(0x008, 1) EDMP_SGB(0x020, 2) EDMP_SGB(0x200, 2)                                       \
                EDMP_SGB(0x008, 1) EDMP_SGB(0x020, 1) EDMP_SGB(0x200, 1)                                       \
                EDMP_SGB(0x008, 1) EDMP_SGB(0x020, 1) EDMP_SGB(0x200, 1)                                       \
                EDMP_SGB(0x008, 1) EDMP_SGB(0x020, 1) EDMP_SGB(0x200, 1)                                       \
            } else {                                                                                           \
                EDMP_SGB(0x100, 2) EDMP_SGB(0x008, 4)                                                          \
            }                                                                                                  \
        }                                                                                                      \
    }
    if constexpr (SPK) {
        constexpr int NGRP = Cf::sk_groups();
        static_assert(NGRP >= 4, "the first four groups carry the staging traffic");
        int kk = 0;
        for (; kk + 1 < nK; kk += 2) {
            EDMP_SK_STEP(lds, P, kk + 2, Q, lds + STAGE)
            __syncthreads();
            EDMP_SK_STEP(lds + STAGE, Q, kk + 3, P, lds)
            __syncthreads();
        }
        if (kk < nK) {  // odd chunk count: the last chunk sits in stage 0 (its fetch / commit are harmless repeats)
            EDMP_SK_STEP(lds, P, kk + 2, Q, lds + STAGE)
            __syncthreads();
        }
    } else {
        int kk = 0;
        for (; kk + 1 < nK; kk += 2) {
            EDMP_RCB_STEP(lds, P, kk + 2, Q, lds + STAGE)
            __syncthreads();
            EDMP_RCB_STEP(lds + STAGE, Q, kk + 3, P, lds)
            __syncthreads();
        }
        if (kk < nK) {  // odd chunk count: the last chunk sits in stage 0
            EDMP_RCB_COMPUTE(lds)
            __syncthreads();
        }
    }
#undef EDMP_SK_STEP
#undef EDMP_SK_MEMSEL
#undef EDMP_RCB_STEP
#undef EDMP_RCB_TILE_SETUP
#undef EDMP_RCB_TAPS
#undef EDMP_RCB_QMEM
#undef EDMP_RCB_MEM0
#undef EDMP_RCB_MEM1
#undef EDMP_RCB_MEM2
#undef EDMP_RCB_MEM3
#undef EDMP_RCB_MFMA4
#undef EDMP_SGB
#undef EDMP_RCB_COMPUTE
#undef EDMP_RCB_RESID
#undef EDMP_RCB_FETCH
#undef EDMP_RCB_COMMIT
#undef EDMP_LD_AP
#undef EDMP_LD_BP
#undef EDMP_ST_AP
#undef EDMP_ST_BP
#undef EDMP_LD_AQ
#undef EDMP_LD_BQ
#undef EDMP_ST_AQ
#undef EDMP_ST_BQ
#undef EDMP_DECL_RA
#undef EDMP_DECL_RB
#undef EDMP_REP7
#undef EDMP_REP10
    __syncthreads();
    EDMP_STAMP(SK, 2)

    // ---- epilogue: raw tile (+bias) -> LDS, per-sample statistics over the whole group, normalise, Mish, add, store
    float* Y = lds;  // [32][YS]; all MFMA reads of the stages are complete (barrier above)
    // this thread's epilogue elements are known up front: request their affine parameters and addends from global memory
    // NOW, so the loads fly under the accumulator spill + statistics phases instead of stalling the final loop
    const int erow = tid >> 3, epart = tid & 7;
    const int eb = min(b0 + erow, p.B - 1);
    float4 g4[NF4], be4[NF4], ad4[NF4];
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
        const int col = 4 * (epart + 8 * i);
        const int l = col / CG, ch = co0 + col % CG;
        g4[i] = *reinterpret_cast<const float4*>(p.gamma + ch);
        be4[i] = *reinterpret_cast<const float4*>(p.beta + ch);
        // exactly one addend per launch (conv1: time bias, conv2: residual; checked by the launcher).  Summing two
        // loads here would put an s_waitcnt vmcnt(0) - a full memory round trip - into every iteration of this loop
        ad4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.add_res) ad4[i] = *reinterpret_cast<const float4*>(p.add_res + ((size_t)eb * L + l) * p.Cout + ch);
        else if (p.add_tb) ad4[i] = *reinterpret_cast<const float4*>(p.add_tb + ch);
    }
    if constexpr (RES) {
        // residual conv (+ its bias): staged through the Y tile and written as float4 by the same (row, 8-column-part)
        // mapping as the final pass - sixteen dword stores per lane straight from the accumulators are store-issue-bound
        if constexpr (SPK) {  // every wave holds a K-slice partial of every tile of its slab: partial buffer wave / S, bias in partial 0
            float* Yw = Y + (wave / S) * (32 * YS);
            const float rb = (wave / S == 0) ? rbias_t0 : 0.0f;
#pragma unroll
            for (int l = 0; l < L; ++l)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    Yw[row * YS + l * CG + (wave % S) * 32 + (lane & 31)] = racc[l][r] + rb;
                }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j = wave + 4 * t;
                if (j < NTILE) {
                    const int l = j / S, s = j % S;
                    const int cc = s * 32 + (lane & 31);
                    const float rb = (t == 0) ? rbias_t0 : rbias_t1;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        Y[row * YS + l * CG + cc] = racc[t][r] + rb;
                    }
                }
            }
        }
        __syncthreads();
        if (b0 + erow < p.B) {
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
    if constexpr (SPK) {
        float* Yw = Y + (wave / S) * (32 * YS);
        const float bias = (wave / S == 0) ? bias_t0 : 0.0f;
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                Yw[row * YS + l * CG + (wave % S) * 32 + (lane & 31)] = acc[l][r] + bias;
            }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = wave + 4 * t;
            if (j < NTILE) {
                const int l = j / S, s = j % S;
                const int cc = s * 32 + (lane & 31);
                const float bias = (t == 0) ? bias_t0 : bias_t1;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    Y[row * YS + l * CG + cc] = acc[t][r] + bias;
                }
            }
        }
    }
    __syncthreads();
    EDMP_STAMP(SK, 3)
    {
        const int row = erow, part = epart;
        const int b = b0 + row;
        float4 v[NF4];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            v[i] = *reinterpret_cast<const float4*>(Y + row * YS + 4 * (part + 8 * i));
#pragma unroll
            for (int q = 1; q < NP; ++q) {  // split-K: the tile is the sum of the four waves' partial tiles
                const float4 pv = *reinterpret_cast<const float4*>(Y + q * (32 * YS) + row * YS + 4 * (part + 8 * i));
                v[i].x += pv.x, v[i].y += pv.y, v[i].z += pv.z, v[i].w += pv.w;
            }
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        sum += __shfl_xor(sum, 4, 64);
        constexpr float inv_n = 1.0f / (float)(L * CG);
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
                const int col = 4 * (part + 8 * i);
                const int l = col / CG, cc = col % CG;
                const int ch = co0 + cc;
                float4 o;
                {
                    const float s0 = rstd * g4[i].x, s1 = rstd * g4[i].y, s2 = rstd * g4[i].z, s3 = rstd * g4[i].w;
                    o.x = mish_fast(v[i].x * s0 + (be4[i].x - s0 * mean)) + ad4[i].x;
                    o.y = mish_fast(v[i].y * s1 + (be4[i].y - s1 * mean)) + ad4[i].y;
                    o.z = mish_fast(v[i].z * s2 + (be4[i].z - s2 * mean)) + ad4[i].z;
                    o.w = mish_fast(v[i].w * s3 + (be4[i].w - s3 * mean)) + ad4[i].w;
                }
                *reinterpret_cast<float4*>(p.dst + ((size_t)b * L + l) * p.Cout + ch) = o;
            }
        }
    }
    EDMP_STAMP(SK, 4)
}


template <int CG, int L, bool RES>
static int launch_rcb_t(const RcbP& p, hipStream_t s) {
    static bool attr_set = false;
    constexpr size_t bytes = RcbCfg<CG, L, RES>::lds_bytes();
    static_assert(bytes <= 160 * 1024, "fused conv kernel exceeds the 160 KiB LDS of a CU");
    if (!attr_set) {
        EDMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rcb_conv_kernel<CG, L, RES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        attr_set = true;
    }
    dim3 grid(p.Cout / CG, (p.B + 31) / 32);
    hipLaunchKernelGGL((rcb_conv_kernel<CG, L, RES>), grid, dim3(256), bytes, s, p);
    return EDMP_OK;
}
}  // namespace edmp
