// levelbench.hip — micro-harness for ONE level_kernel instance with dummy operands (timing only): microseconds per launch and the
// phase stamps of workgroup 0.  hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DMODE=LV_UP -DCV=64 -DLV=13 -DCINV=256 -DSLOT=3]
// tools/levelbench.hip -o tools/levelbench
#define EDMP_STAMPS 1
#define EDMP_STAMPS_DEFINE 1
#include "../edmp_amd/csrc/common.h"
#include "../edmp_amd/csrc/params.h"
#include <cstdio>
#include <vector>
#include <random>
namespace edmp { void set_error(const char*, ...) {} }
#include "../edmp_amd/csrc/wide.hip"
#include "../edmp_amd/csrc/level.hip"
using namespace edmp;
#ifndef MODE
#define MODE LV_UP_FINAL
#define CV 32
#define LV 25
#define CINV 128
#define SLOT 4
#endif
int main() {
    const int B = 1024;
    std::mt19937 g(1);
    std::uniform_real_distribution<float> d(-1.f, 1.f);
    const size_t n = 16u << 20;
    std::vector<float> h(n);
    for (auto& v : h) v = d(g) * 0.05f;
    float *buf, *out, *skip;
    hipMalloc((void**)&buf, n * 4); hipMalloc((void**)&out, n * 4); hipMalloc((void**)&skip, n * 4);
    hipMemcpy(buf, h.data(), n * 4, hipMemcpyHostToDevice);
    LevelP p{};
    const float** pp = reinterpret_cast<const float**>(&p);
    p.src1 = buf; p.src2 = (MODE == LV_DOWN) ? nullptr : buf + (4u << 20);
    p.C1 = (MODE == LV_DOWN) ? CINV : CINV / 2; p.C2 = (MODE == LV_DOWN) ? 0 : CINV / 2;
    p.w11 = buf; p.w12 = buf + 1000; p.w21 = buf + 2000; p.w22 = buf + 3000; p.wrs = buf + 4000; p.wfin = buf + 5000;
    p.b11 = p.g11 = p.be11 = p.tb1 = p.rb1 = p.b12 = p.g12 = p.be12 = p.b21 = p.g21 = p.be21 = p.tb2 = p.b22 = p.g22 = p.be22 = p.brs = p.bfin = p.gfin = p.befin = buf + 7000;
    p.skip_out = (MODE == LV_DOWN) ? skip : nullptr; p.out = out; p.B = B;
    (void)pp;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch_level_t<MODE, CV, LV, 4, CINV>(p, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 50; ++i) launch_level_t<MODE, CV, LV, 4, CINV>(p, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long st[8][16];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(edmp::g_stamps), sizeof(st));
    printf("%.2f us/launch (in-kernel %.2f us) | zero+load %llu | conv11 %llu | epi11 %llu | conv12 %llu | epi12 %llu | RCB2 %llu | resample %llu | %s\n", ms * 1000 / 50, (st[SLOT][15] - st[SLOT][1]) / 100.0,
           st[SLOT][2] - st[SLOT][0], st[SLOT][4] - st[SLOT][2], st[SLOT][6] - st[SLOT][4], st[SLOT][8] - st[SLOT][6], st[SLOT][10] - st[SLOT][8], st[SLOT][12] - st[SLOT][10], st[SLOT][14] - st[SLOT][12], hipGetErrorString(hipGetLastError()));
    return 0;
}
