// mfma4x4_probe.hip — lane layout and issue rate of v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 blocks, K = 1), the instruction the level
// kernels use for a row tile with <= 4 real rows (level.hip, round 5).  Assumed layout, checked here against a host product: lane l feeds
// A[block l/4][row l%4] and B[block l/4][col l%4]; VGPR r of lane l returns D[block l/4][row r][col l%4].  Also: shader cycles per instruction
// in a loop over 4 independent accumulators (8 = 2 passes: the rate of the 16x16x4 instruction, 64 FLOP / cycle / SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma4x4_probe.hip -o tools/mfma4x4_probe && tools/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using f4 = __attribute__((ext_vector_type(4))) float;
__device__ unsigned long long g_cyc;
__global__ void k(const float* A, const float* B, float* D, int iters) {
    const int l = threadIdx.x;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
    f4 acc[4] = {c, c, c, c};
    const float a = A[l], b = B[l];
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[u & 3], 0, 0, 0);
    }
    const unsigned long long t1 = clock64();
    D[256 + l] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (l == 0) g_cyc = t1 - t0;
}
int main() {
    float hA[64], hB[64], hD[512];
    for (int i = 0; i < 64; ++i) hA[i] = 1.0f + i, hB[i] = 0.5f + 0.25f * i;
    float *A, *B, *D;
    hipMalloc(&A, 256), hipMalloc(&B, 256), hipMalloc(&D, 2048);
    hipMemcpy(A, hA, 256, hipMemcpyHostToDevice), hipMemcpy(B, hB, 256, hipMemcpyHostToDevice);
    const int iters = 100000;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, D, iters);
    hipMemcpy(hD, D, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const float want = hA[(l / 4) * 4 + r] * hB[l];  // A[block][row r] * B[block][col l%4]
            if (hD[l * 4 + r] != want) ++bad;
        }
    unsigned long long cyc;
    hipMemcpyFromSymbol(&cyc, HIP_SYMBOL(g_cyc), sizeof(cyc));
    printf("layout check: %d of 256 elements differ from D[blk][r][c] = A[blk][r] * B[blk][c] at (lane 4 blk + c, VGPR r)\n", bad);
    printf("issue: %.2f shader cycles per v_mfma_f32_4x4x1_16B_f32 (4 independent accumulators)\n", (double)cyc / (iters * 8.0));
    return 0;
}
