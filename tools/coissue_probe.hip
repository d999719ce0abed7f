// coissue_probe.hip — can ONE wave keep the matrix pipe busy while it issues VALU work between its MFMAs?  (the question behind
// software-pipelining the level kernels' GroupNorm / Mish epilogues under the next tiles' MFMAs)
//   hipcc --offload-arch=gfx950 -O3 tools/coissue_probe.hip -o tools/coissue_probe && tools/coissue_probe
// Loop of v_mfma_f32_16x16x4_f32 (8 passes = 32 shader cycles each, 2 independent accumulator chains) with NV independent VALU
// instructions (v_fma_f32 on private registers; or v_exp_f32 = a quarter-rate transcendental) issued behind every MFMA, one wave
// per SIMD on all 256 CUs.  Printed: shader cycles per MFMA - 32.0 means the VALU work was free.
#include <hip/hip_runtime.h>

#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ unsigned long long g_st[2];

template <int NV, int TRANS>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    f32x4 c[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 1.0f + 1e-3f * (threadIdx.x + i);
    const float a = 1.f + threadIdx.x * 1e-3f, b = 2.f + threadIdx.x * 2e-3f, k = 0.999f, d = 1e-4f;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            c[u & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[u & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 7]));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(k), "v"(d));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + c[0][0] + c[1][1];
    if (threadIdx.x == 0 && blockIdx.x == 7) g_st[0] = t1 - t0;
}

template <int NV, int TRANS>
static void run(float* d) {
    const int iters = 20000;
    hipLaunchKernelGGL((probe<NV, TRANS>), dim3(256), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((probe<NV, TRANS>), dim3(256), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    unsigned long long st[2];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_st), sizeof(st));
    printf("%s x %2d behind every v_mfma_f32_16x16x4_f32: %6.2f shader cycles per MFMA\n", TRANS ? "v_exp_f32" : "v_fma_f32", NV, (double)st[0] / (iters * 8.0));
}

int main() {
    float* d;
    hipMalloc(&d, 1 << 20);
    run<0, 0>(d); run<1, 0>(d); run<2, 0>(d); run<3, 0>(d); run<4, 0>(d); run<5, 0>(d); run<6, 0>(d); run<7, 0>(d); run<8, 0>(d); run<12, 0>(d); run<16, 0>(d);
    run<1, 1>(d); run<2, 1>(d); run<4, 1>(d);
    hipFree(d);
    return 0;
}
