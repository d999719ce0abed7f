// coissue_control.hip — the CONTROL the round-4 review asked for (VERDICT r4 "the co-issue conclusion has no control"):
// tools/coissue_probe.hip found that ONE VALU instruction behind a v_mfma_f32_16x16x4_f32 costs the issuing wave 13.5 cycles and
// concluded that a wave cannot hide VALU work under its own fp32 MFMAs.  The microarchitecture guide measures the opposite for the
// bf16 pipe (<= 5 single-issue fillers hidden per 32-cycle v_mfma_f32_32x32x16_bf16 gap from one wave).  Same harness, same
// fillers (inline-asm v_fma_f32 on private registers, fenced by sched_barrier(0)), four matrix instructions:
//     K0  v_mfma_f32_16x16x4_f32   (8 passes, 32 cycles)       K1  v_mfma_f32_32x32x2_f32   (16 passes, 64 cycles)
//     K2  v_mfma_f32_32x32x16_bf16 (8 passes, 32 cycles)       K3  v_mfma_f32_16x16x32_bf16 (4 passes, 16 cycles)
// If K2 shows ~32 cycles per MFMA with up to ~5 fillers, the harness reproduces the known-good case and the fp32 rows are a fact
// about the fp32-input MFMA (it shares the issue / operand path of the fp32 vector unit), not about the harness.
//   hipcc --offload-arch=gfx950 -O3 tools/coissue_control.hip -o tools/coissue_control && tools/coissue_control
// One wave per SIMD on all 256 CUs, NCH independent accumulator chains (2 or 4), NV fillers behind EVERY MFMA.
#include <hip/hip_runtime.h>

#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ unsigned long long g_st[2];

template <int KIND>
struct Acc {
    using type = f32x16;
    static constexpr int N = 16;
};
template <>
struct Acc<0> {
    using type = f32x4;
    static constexpr int N = 4;
};
template <>
struct Acc<3> {
    using type = f32x4;
    static constexpr int N = 4;
};

template <int KIND, int NV, int NCH>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    using acc_t = typename Acc<KIND>::type;
    acc_t c[NCH];
#pragma unroll
    for (int n = 0; n < NCH; ++n)
#pragma unroll
        for (int i = 0; i < Acc<KIND>::N; ++i) c[n][i] = 0.f;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 1.0f + 1e-3f * (threadIdx.x + i);
    const float a = 1.f + threadIdx.x * 1e-3f, b = 2.f + threadIdx.x * 2e-3f, k = 0.999f, d = 1e-4f;
    u32x4_t ua, ub;
#pragma unroll
    for (int i = 0; i < 4; ++i) ua[i] = 0x3c003c00u | (threadIdx.x * 7 + i), ub[i] = 0x3c003c00u | (threadIdx.x * 3 + i);
    const bf16x8_t ba = __builtin_bit_cast(bf16x8_t, ua), bb = __builtin_bit_cast(bf16x8_t, ub);
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            constexpr int dummy = 0;
            (void)dummy;
            if constexpr (KIND == 0) c[u % NCH] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[u % NCH], 0, 0, 0);
            else if constexpr (KIND == 1) c[u % NCH] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[u % NCH], 0, 0, 0);
            else if constexpr (KIND == 2) c[u % NCH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, c[u % NCH], 0, 0, 0);
            else c[u % NCH] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, c[u % NCH], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(k), "v"(d));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
    for (int n = 0; n < NCH; ++n) s += c[n][0] + c[n][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 7) g_st[0] = t1 - t0;
}

template <int KIND, int NV, int NCH>
static double run(float* d) {
    const int iters = 20000;
    hipLaunchKernelGGL((probe<KIND, NV, NCH>), dim3(256), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((probe<KIND, NV, NCH>), dim3(256), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    unsigned long long st[2];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_st), sizeof(st));
    return (double)st[0] / (iters * 8.0);
}

template <int KIND, int NCH>
static void row(float* d, const char* name) {
    printf("| %s, %d chains |", name, NCH);
    printf(" %.1f |", run<KIND, 0, NCH>(d));
    printf(" %.1f |", run<KIND, 1, NCH>(d));
    printf(" %.1f |", run<KIND, 2, NCH>(d));
    printf(" %.1f |", run<KIND, 3, NCH>(d));
    printf(" %.1f |", run<KIND, 4, NCH>(d));
    printf(" %.1f |", run<KIND, 5, NCH>(d));
    printf(" %.1f |", run<KIND, 6, NCH>(d));
    printf(" %.1f |", run<KIND, 8, NCH>(d));
    printf(" %.1f |", run<KIND, 12, NCH>(d));
    printf(" %.1f |\n", run<KIND, 16, NCH>(d));
    fflush(stdout);
}

int main() {
    float* d;
    hipMalloc(&d, 1 << 20);
    printf("shader cycles per MFMA with N x v_fma_f32 issued behind every MFMA by the same wave (one wave per SIMD, 256 CUs)\n");
    printf("| matrix instruction | N = 0 | 1 | 2 | 3 | 4 | 5 | 6 | 8 | 12 | 16 |\n|---|---|---|---|---|---|---|---|---|---|---|\n");
    row<2, 2>(d, "v_mfma_f32_32x32x16_bf16 (32 cyc)");
    row<2, 4>(d, "v_mfma_f32_32x32x16_bf16 (32 cyc)");
    row<3, 4>(d, "v_mfma_f32_16x16x32_bf16 (16 cyc)");
    row<0, 2>(d, "v_mfma_f32_16x16x4_f32 (32 cyc)");
    row<0, 4>(d, "v_mfma_f32_16x16x4_f32 (32 cyc)");
    row<1, 2>(d, "v_mfma_f32_32x32x2_f32 (64 cyc)");
    row<1, 4>(d, "v_mfma_f32_32x32x2_f32 (64 cyc)");
    hipFree(d);
    return 0;
}
