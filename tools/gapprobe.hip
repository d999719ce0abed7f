// gapprobe.hip — back-to-back launch cost of trivial kernels on one stream (the floor under every dependent launch of the layer
// program).  hipcc --offload-arch=gfx950 -O3 tools/gapprobe.hip -o tools/gapprobe.  MI355X, ROCm 7.0: 2.6 us (1 workgroup), 2.7 us
// (256 x 256 threads), 2.8 us (+120 KB LDS), 3.1 / 4.6 / 6.8 us when the kernel also stores 4 / 16 / 32 MB; the same chain
// replayed from a hipGraph: 1.55 us per node (host enqueue cost removed - real kernel chains do not get faster).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_k(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }
__global__ __launch_bounds__(256) void lds_k(float* p) { extern __shared__ float l[]; if (p && threadIdx.x == 9999) { l[0] = 1; p[0] = l[1]; } }
__global__ __launch_bounds__(256) void store_k(float* p, int n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; for (int k = 0; k < n; ++k) p[i + (size_t)k * 65536] = 1.0f; }
template <class F> float timeit(F f, int n) { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); for (int i = 0; i < 20; ++i) f(); hipEventRecord(a, 0); for (int i = 0; i < n; ++i) f(); hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000 / n; }
int main() {
    float* p; hipMalloc((void**)&p, 256u << 20);
    hipFuncSetAttribute((const void*)lds_k, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    printf("empty 1 block:            %.2f us/launch\n", timeit([&] { hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, 0, p); }, 2000));
    printf("empty 256 x 256:          %.2f us/launch\n", timeit([&] { hipLaunchKernelGGL(empty_k, dim3(256), dim3(256), 0, 0, p); }, 2000));
    printf("256 x 256, 120 KB LDS:    %.2f us/launch\n", timeit([&] { hipLaunchKernelGGL(lds_k, dim3(256), dim3(256), 120 * 1024, 0, p); }, 2000));
    for (int n : {1, 16, 64, 128}) printf("256 x 256 storing %3d MB:  %.2f us/launch\n", n * 65536 * 4 / 1048576, timeit([&] { hipLaunchKernelGGL(store_k, dim3(256), dim3(256), 0, 0, p, n); }, 500));
    {  // the same chain as one hipGraph launch: host enqueue cost out of the picture
        hipStream_t st; hipStreamCreate(&st);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(lds_k, dim3(256), dim3(256), 120 * 1024, st, p);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipGraphLaunch(ge, st); hipStreamSynchronize(st);
        hipEventRecord(a, st); hipGraphLaunch(ge, st); hipEventRecord(b, st); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("hipGraph of 1000 x (256 x 256, 120 KB LDS): %.2f us/launch\n", ms);
    }
    return 0;
}
