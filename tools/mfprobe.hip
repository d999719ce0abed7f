// mfprobe.hip — what the fp32 matrix pipe of THIS box sustains (the denominator question of DESIGN.md §5).
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfprobe.hip -o tools/mfprobe && tools/mfprobe
//
// Pure MFMA loops, no memory traffic: v_mfma_f32_32x32x2_f32 (16 passes, 64 shader cycles per SIMD, 2 x 32 x 32 x 2 flop)
// and v_mfma_f32_16x16x4_f32 (32 cycles per SIMD) with 1 / 2 / 4 independent accumulator chains per wave, one or two
// waves per SIMD, on all 256 CUs, for launch durations from ~50 us to ~200 ms.  Two clocks are read: s_memtime (shader
// cycles, what the kernel itself sees) and the 100 MHz wall clock; the host times the launch with HIP events.  Printed
// per run: MFMAs per wave, shader cycles per MFMA, the shader clock implied by the two counters, ns per MFMA, and
// TFLOP/s = flops of the whole grid / event time, next to the 157.3 TFLOP/s spec peak (256 CUs x 4 SIMDs x 64 flop/clk x
// 2.4 GHz).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ unsigned long long g_st[2];

template <int SHAPE, int NACC>
__global__ __launch_bounds__(512) void probe(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    f32x4 c[NACC];
#pragma unroll
    for (int n = 0; n < NACC; ++n) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) c[n][i] = 0.f;
    }
    const float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int n = 0; n < NACC; ++n) {
                if (SHAPE == 32) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
                else c[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[n], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < NACC; ++n) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[n][i];
#pragma unroll
        for (int i = 0; i < 4; ++i) s += c[n][i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 7) {
        g_st[0] = t1 - t0;
        g_st[1] = w1 - w0;
    }
}

template <int SHAPE, int NACC>
static void run(int threads, int iters, float* d) {
    const int blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<SHAPE, NACC>), dim3(blocks), dim3(threads), 0, 0, d, iters, 1.f, 2.f);  // warm
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<SHAPE, NACC>), dim3(blocks), dim3(threads), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long st[2];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_st), sizeof(st));
    const double per_wave = (double)iters * 8 * NACC;
    const double flop_each = SHAPE == 32 ? 2.0 * 32 * 32 * 2 : 2.0 * 16 * 16 * 4;
    const double waves = (double)blocks * threads / 64;
    const double tf = per_wave * flop_each * waves / (ms * 1e-3) / 1e12;
    printf("mfma_f32_%s  chains/wave %d  waves/SIMD %d  %9.0f mfma/wave  %7.2f shader cyc/mfma/wave  clock %.3f GHz  %6.2f ns/mfma/wave  event %9.3f ms  %6.1f TFLOP/s = %.3f of 157.3\n",
           SHAPE == 32 ? "32x32x2 " : "16x16x4 ", NACC, threads / 256, per_wave, (double)st[0] / per_wave, st[0] / (st[1] * 10.0), st[1] * 10.0 / per_wave, ms, tf,
           tf / 157.3);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main() {
    float* d;
    hipMalloc(&d, 1 << 22);
    for (int iters : {256, 4096, 65536, 1048576}) {
        run<32, 1>(256, iters, d);
        run<32, 2>(256, iters / 2, d);
        run<32, 4>(256, iters / 4, d);
        run<32, 1>(512, iters / 2, d);
        run<32, 2>(512, iters / 4, d);
        run<16, 1>(256, iters * 2, d);
        run<16, 2>(256, iters, d);
        run<16, 4>(256, iters / 2, d);
        run<16, 2>(512, iters / 2, d);
        printf("\n");
    }
    hipFree(d);
    return 0;
}
