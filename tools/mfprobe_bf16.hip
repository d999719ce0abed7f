// mfprobe_bf16.hip — what the bf16 matrix pipe of THIS box sustains, next to the fp32 pipe (tools/mfprobe.hip): the question
// behind the bf16x3 exact-product experiment (profiles/r04_bf16x3_l4_l7.md) - nine v_mfma_f32_32x32x16_bf16 (8 passes, 32 shader
// cycles) replace eight v_mfma_f32_32x32x2_f32 (16 passes, 64 cycles) only if the chip holds its clock under the denser pipe.
//   hipcc --offload-arch=gfx950 -O3 tools/mfprobe_bf16.hip -o tools/mfprobe_bf16 && tools/mfprobe_bf16
// Pure MFMA loops on all 256 CUs, one wave per SIMD, 2 independent accumulator chains, operands with random mantissas (data
// toggling matters for power); s_memtime (shader cycles) against the 100 MHz wall clock gives the clock the kernel ran at.
#include <hip/hip_runtime.h>

#include <cstdio>

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__device__ unsigned long long g_st[2];

template <int KIND>  // 0: fp32 32x32x2, 1: bf16 32x32x16, 2: the 9 : 8 mix has no meaning here - see the kernels
__global__ __launch_bounds__(256) void probe(float* out, int iters, unsigned seed) {
    f32x16 acc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    unsigned h = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return h; };
    // bf16 operands: random mantissas, exponents near 1.0 so that nothing overflows over millions of accumulations (sums stay finite: |x| < 2, alternating signs)
    u32x4_t ua, ub;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ua[i] = (rnd() & 0x807f807fu) | 0x3c003c00u;
        ub[i] = (rnd() & 0x807f807fu) | 0x3c003c00u;
    }
    const float fa = __builtin_bit_cast(float, (rnd() & 0x807fffffu) | 0x3c000000u), fb = __builtin_bit_cast(float, (rnd() & 0x807fffffu) | 0x3c000000u);
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                if (KIND == 0) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[n], 0, 0, 0);
                else acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ua), __builtin_bit_cast(bf16x8_t, ub), acc[n], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 7) {
        g_st[0] = t1 - t0;
        g_st[1] = w1 - w0;
    }
}

template <int KIND>
static void run(int iters, float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<KIND>), dim3(256), dim3(256), 0, 0, d, iters, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<KIND>), dim3(256), dim3(256), 0, 0, d, iters, 2u);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long st[2];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_st), sizeof(st));
    const double per_wave = (double)iters * 16;
    const double flop_each = KIND == 0 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16;
    const double tf = per_wave * flop_each * 1024 / (ms * 1e-3) / 1e12;
    printf("%s  %9.0f mfma/wave  %6.2f shader cyc/mfma  clock %.3f GHz  %6.2f ns/mfma  event %8.3f ms  %7.1f TFLOP/s (spec %s)\n", KIND == 0 ? "mfma_f32_32x32x2_f32  " : "mfma_f32_32x32x16_bf16",
           per_wave, (double)st[0] / per_wave, st[0] / (st[1] * 10.0), st[1] * 10.0 / per_wave, ms, tf, KIND == 0 ? "157.3" : "2516.6");
}

int main() {
    float* d;
    hipMalloc(&d, 1 << 22);
    for (int rep = 0; rep < 2; ++rep)
        for (int iters : {4096, 65536, 524288}) {
            run<0>(iters / 2, d);
            run<1>(iters, d);
        }
    hipFree(d);
    return 0;
}
