// widebench.hip — micro-harness for ONE wide_conv_kernel instance (compiles in seconds; kbench covers all of them): 50 launches
// at B = 1024 on random data, microseconds per launch and the s_memtime phase stamps of workgroup 0 (prologue | K loop and the
// part of it spent at the chunk barrier | spill | statistics + store).  Timing tool, no reference check.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DKIND=WK_K5K4 -DMSV=32 -DCGV=64 -DGSV=64 -DLV=4] tools/widebench.hip -o tools/widebench
#define EDMP_STAMPS 1
#define EDMP_STAMPS_DEFINE 1
#include "../edmp_amd/csrc/common.h"
#include "../edmp_amd/csrc/params.h"
#include <type_traits>
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <random>
namespace edmp {
void set_error(const char*, ...) {}

}
#ifndef EDMP_EXP
#define EDMP_EXP 0
#endif
#include "../edmp_amd/csrc/wide.hip"
using namespace edmp;
#ifndef KIND
#define KIND WK_K5K2
#define MSV 32
#define CGV 64
#define GSV 64
#define LV 2
#endif
int main() {
    const int B = 1024, C = GSV * 8, L = LV;
    using Cf = WideCfg<KIND, MSV, CGV, GSV, LV, false>;
    std::mt19937 g(1);
    std::uniform_real_distribution<float> d(-1.f, 1.f);
    std::vector<float> hx((size_t)B * L * C), hw((size_t)(C / Cf::SW) * (C / Cf::KG) * Cf::NSLAB * 256), hp(C);
    for (auto& v : hx) v = d(g);
    for (auto& v : hw) v = d(g) * 0.02f;
    for (auto& v : hp) v = d(g);
    float *x, *w, *pp, *y;
    hipMalloc((void**)&x, hx.size() * 4); hipMalloc((void**)&w, hw.size() * 4); hipMalloc((void**)&pp, C * 4); hipMalloc((void**)&y, hx.size() * 4);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice); hipMemcpy(pp, hp.data(), C * 4, hipMemcpyHostToDevice);
    RcbP p{};
    p.src1 = x; p.C1 = C; p.W = w; p.bias = pp; p.gamma = pp; p.beta = pp; p.add_tb = pp; p.dst = y; p.Cout = C; p.B = B;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch_wide_t<KIND, MSV, CGV, GSV, LV, false>(p, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 50; ++i) launch_wide_t<KIND, MSV, CGV, GSV, LV, false>(p, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long st[8][16];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(edmp::g_stamps), sizeof(st));
    std::vector<float> hy(8); hipMemcpy(hy.data(), y, 32, hipMemcpyDeviceToHost);
    printf("%.2f us/launch (in-kernel %.2f us wall) | prologue %llu | K loop %llu (barrier %llu) | spill %llu | stats+store %llu | y0 %g\n", ms * 1000 / 50, (st[0][9] - st[0][1]) / 100.0, st[0][2] - st[0][0], st[5][4], st[5][0], st[0][6] - st[0][4], st[0][8] - st[0][6], hy[0]);
    return 0;
}
