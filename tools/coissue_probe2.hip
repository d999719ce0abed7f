// coissue_probe2.hip — do TWO waves on one SIMD overlap matrix and vector work?  512 threads per workgroup = 2 waves per SIMD on
// all 256 CUs: waves 0-3 run a pure v_mfma_f32_16x16x4_f32 loop (32 shader cycles each), waves 4-7 a pure VALU loop (v_fma_f32 or
// v_exp_f32), each alone and both together.  Printed: shader cycles per MFMA of the MFMA waves and per VALU instruction of the VALU waves.
//   hipcc --offload-arch=gfx950 -O3 tools/coissue_probe2.hip -o tools/coissue_probe2 && tools/coissue_probe2
#include <hip/hip_runtime.h>

#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ unsigned long long g_st[2];

template <int TRANS>
__global__ __launch_bounds__(512) void probe(float* out, int iters_m, int iters_v) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4) {
        f32x4 c[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const float a = 1.f + threadIdx.x * 1e-3f, b = 2.f + threadIdx.x * 2e-3f;
        const unsigned long long t0 = clock64();
        for (int it = 0; it < iters_m; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) c[u & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[u & 1], 0, 0, 0);
        }
        const unsigned long long t1 = clock64();
        s = c[0][0] + c[1][1];
        if (threadIdx.x == 0 && blockIdx.x == 7) g_st[0] = t1 - t0;
    } else {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 1.0f + 1e-3f * (threadIdx.x + i);
        const float k = 0.999f, d = 1e-4f;
        const unsigned long long t0 = clock64();
        for (int it = 0; it < iters_v; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 7]));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(k), "v"(d));
            }
        }
        const unsigned long long t1 = clock64();
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
        if (threadIdx.x == 256 && blockIdx.x == 7) g_st[1] = t1 - t0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int TRANS>
static void run(float* d, int im, int iv) {
    unsigned long long z[2] = {0, 0}, st[2];
    hipMemcpyToSymbol(HIP_SYMBOL(g_st), z, sizeof(z));
    hipLaunchKernelGGL((probe<TRANS>), dim3(256), dim3(512), 0, 0, d, im, iv);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((probe<TRANS>), dim3(256), dim3(512), 0, 0, d, im, iv);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_st), sizeof(st));
    printf("%s | MFMA waves: %s  VALU waves: %s", TRANS ? "v_exp_f32" : "v_fma_f32", im ? "on " : "off", iv ? "on " : "off");
    if (im) printf(" | %6.2f shader cycles per MFMA", (double)st[0] / (im * 8.0));
    if (iv) printf(" | %6.2f shader cycles per VALU instruction", (double)st[1] / (iv * 16.0));
    printf("\n");
}

int main() {
    float* d;
    hipMalloc(&d, 1 << 21);
    // the VALU loop is sized to last about as long as the MFMA loop when both run
    run<0>(d, 20000, 0);
    run<0>(d, 0, 40000);
    run<0>(d, 20000, 80000);
    run<0>(d, 20000, 40000);
    run<1>(d, 0, 10000);
    run<1>(d, 20000, 20000);
    hipFree(d);
    return 0;
}
