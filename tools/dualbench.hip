// dualbench.hip — VERDICT r4 item 1 in isolation, on the CURRENT kernels: a chain of NL dependent launches of one position-tile
// instance (ping-pong between two activation buffers, the way consecutive Conv1dBlocks of a level follow each other)
//   (a) as the product runs it: ONE stream, 32-sample tiles, B = 1024 rows  -> 256 workgroups per launch, one per CU
//   (b) ONE stream, 16-sample tiles, B = 1024                                -> 512 workgroups per launch, two per CU, in phase
//   (c) TWO streams, 16-sample tiles, 512 rows each (fork - two half-batch chains - join), the second chain delayed by D us
//       -> 2 x 256 workgroups in flight from different launches: one chain's prologue / epilogue / kernel boundary can run under the
//       other chain's K loop
// Timing tool (random data, no reference check).  Printed: us per layer of the whole 1024-row batch for each mode.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=12 [-DKIND=WK_K5 -DCGV=32 -DGSV=32 -DLV=7 [-DCIN=128 -DRESV=true]] tools/dualbench.hip -o tools/dualbench && tools/dualbench [layers = 10] [1: also the two-stream modes]
#include "../edmp_amd/csrc/common.h"
#include "../edmp_amd/csrc/params.h"
#include <type_traits>
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>
namespace edmp {
void set_error(const char*, ...) {}
}
#include "../edmp_amd/csrc/wide.hip"
using namespace edmp;
#ifndef KIND
#define KIND WK_K5
#define CGV 32
#define GSV 32
#define LV 7
#endif
#ifndef CG16
#define CG16 CGV  // channels per workgroup of the 16-sample instance
#endif
#ifndef RESV
#define RESV false  // fold the block's residual 1x1 conv (needs CINV != channels)
#endif
#ifndef CINV
#define CINV (GSV * 8)  // input channels
#endif

__global__ void delay_kernel(long long cycles) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
}

template <int MS, int CG>
static void chain(const RcbP& base, float* x, float* y, int B, int nl, hipStream_t s) {
    for (int i = 0; i < nl; ++i) {
        RcbP p = base;
        p.src1 = (i & 1) ? y : x;
        p.dst = (i & 1) ? x : y;
        p.B = B;
        launch_wide_t<KIND, MS, CG, GSV, LV, RESV>(p, s);
    }
}

int main(int argc, char** argv) {
    const int NL = argc > 1 ? atoi(argv[1]) : 10;
    const int B = 1024, C = GSV * 8, L = LV;
    using C32 = WideCfg<KIND, 32, CGV, GSV, LV, RESV>;
    using C16 = WideCfg<KIND, 16, CG16, GSV, LV, RESV>;
    constexpr int LMAX = C32::LOUT > LV ? C32::LOUT : LV, CMAX = CINV > GSV * 8 ? CINV : GSV * 8;  // (resampling / residual instances: shapes differ, the ping-pong just reuses big enough buffers)
    std::mt19937 g(1);
    std::uniform_real_distribution<float> d(-1.f, 1.f);
    std::vector<float> hx((size_t)B * LMAX * CMAX), hw32((size_t)(C / C32::SW) * (CINV / C32::KG) * C32::NSLAB * 256), hw16((size_t)(C / C16::SW) * (CINV / C16::KG) * C16::NSLAB * 256), hp(C);
    (void)L;
    for (auto& v : hx) v = d(g);
    for (auto& v : hw32) v = d(g) * 0.02f;
    for (auto& v : hw16) v = d(g) * 0.02f;
    for (auto& v : hp) v = d(g);
    float *x, *y, *w32, *w16, *pp, *ro;
    hipMalloc((void**)&x, hx.size() * 4), hipMalloc((void**)&y, hx.size() * 4), hipMalloc((void**)&ro, hx.size() * 4);
    hipMalloc((void**)&w32, hw32.size() * 4), hipMalloc((void**)&w16, hw16.size() * 4), hipMalloc((void**)&pp, C * 4);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice), hipMemcpy(y, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w32, hw32.data(), hw32.size() * 4, hipMemcpyHostToDevice), hipMemcpy(w16, hw16.data(), hw16.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(pp, hp.data(), C * 4, hipMemcpyHostToDevice);
    RcbP p{};
    p.C1 = CINV, p.bias = pp, p.gamma = pp, p.beta = pp, p.add_tb = pp, p.Cout = C;
    if (RESV) p.res_out = ro, p.res_bias = pp;
    hipStream_t s0, s1;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking), hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipEvent_t e0, e1, ef, ej;
    hipEventCreate(&e0), hipEventCreate(&e1), hipEventCreateWithFlags(&ef, hipEventDisableTiming), hipEventCreateWithFlags(&ej, hipEventDisableTiming);
    auto timed = [&](auto&& body) {
        float best = 1e30f;
        for (int r = 0; r < 7; ++r) {
            hipEventRecord(e0, s0);
            body();
            hipEventRecord(e1, s0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (r >= 2) best = std::min(best, ms);
        }
        return best * 1000.0f / NL;
    };
    const size_t half = (size_t)512 * LMAX * CMAX;
    const bool two = argc > 2 && atoi(argv[2]);
    p.W = w32;
    const float t32 = timed([&] { chain<32, CGV>(p, x, y, B, NL, s0); });
    p.W = w16;
    const float t16 = timed([&] { chain<16, CG16>(p, x, y, B, NL, s0); });
    printf("%d layers, instance <%d, *, %d|%d, %d, %d, %s> Cin %d: one stream 32-row tiles %.2f us/layer | one stream 16-row tiles (2 workgroups per CU, in phase) %.2f us/layer (x%.3f)\n", NL,
           (int)KIND, CGV, CG16, GSV, LV, RESV ? "res" : "-", CINV, t32, t16, t32 / t16);
    for (int dus : {0, 3, 6, 10, 15, 20}) {
        if (!two) break;
        const float t2 = timed([&] {
            hipEventRecord(ef, s0);
            hipStreamWaitEvent(s1, ef, 0);
            if (dus) hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, s1, (long long)dus * 2400);
            chain<16, CG16>(p, x, y, 512, NL, s0);
            chain<16, CG16>(p, x + half, y + half, 512, NL, s1);
            hipEventRecord(ej, s1);
            hipStreamWaitEvent(s0, ej, 0);
        });
        printf("  two streams x 512 rows, 16-row tiles, second chain delayed %2d us: %.2f us/layer incl. fork + join (x%.3f vs one stream 32-row)\n", dus, t2, t32 / t2);
    }
    // what the fork + join alone costs: two EMPTY-ish chains (1 layer each)
    if (two) {
        hipEvent_t a, b;
        hipEventCreate(&a), hipEventCreate(&b);
        float best = 1e30f;
        for (int r = 0; r < 7; ++r) {
            hipEventRecord(a, s0);
            for (int k = 0; k < 20; ++k) {
                hipEventRecord(ef, s0);
                hipStreamWaitEvent(s1, ef, 0);
                hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, s0, 0LL);
                hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, s1, 0LL);
                hipEventRecord(ej, s1);
                hipStreamWaitEvent(s0, ej, 0);
            }
            hipEventRecord(b, s0);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (r >= 2) best = std::min(best, ms);
        }
        float best1 = 1e30f;
        for (int r = 0; r < 7; ++r) {
            hipEventRecord(a, s0);
            for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, s0, 0LL);
            hipEventRecord(b, s0);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (r >= 2) best1 = std::min(best1, ms);
        }
        printf("  fork + join around one empty kernel per stream: %.2f us per fork-join (one empty kernel on one stream: %.2f us)\n", best * 1000 / 20, best1 * 1000 / 20);
    }
    return 0;
}
