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
#ifndef SBV
#define SBV 4  // samples per workgroup (2: two co-resident workgroups per CU)
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
    for (int i = 0; i < 5; ++i) launch_level_t<MODE, CV, LV, SBV, CINV>(p, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 50; ++i) launch_level_t<MODE, CV, LV, SBV, CINV>(p, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long st[8][16];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(edmp::g_stamps), sizeof(st));
    printf("%.2f us/launch (in-kernel %.2f us) | zero+load %llu | conv11 %llu | epi11 %llu | conv12 %llu | epi12 %llu | RCB2 %llu | resample %llu | %s\n", ms * 1000 / 50, (st[SLOT][15] - st[SLOT][1]) / 100.0,
           st[SLOT][2] - st[SLOT][0], st[SLOT][4] - st[SLOT][2], st[SLOT][6] - st[SLOT][4], st[SLOT][8] - st[SLOT][6], st[SLOT][10] - st[SLOT][8], st[SLOT][12] - st[SLOT][10], st[SLOT][14] - st[SLOT][12], hipGetErrorString(hipGetLastError()));
    {   // dispatch timeline of ONE launch (the last of the chain): when every workgroup started / ended, and on which CU
        const int nwg = (B + SBV - 1) / SBV;
        static unsigned long long wt[1024][4];
        hipMemcpyFromSymbol(wt, HIP_SYMBOL(edmp::g_wg_times), sizeof(wt));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < nwg && i < 1024; ++i) { t0 = wt[i][0] < t0 ? wt[i][0] : t0; t1 = wt[i][1] > t1 ? wt[i][1] : t1; }
        int hist[64] = {0};
        double dur = 0, late = 0;
        std::vector<int> percu(8 * 64, 0);
        for (int i = 0; i < nwg && i < 1024; ++i) {
            const double st0 = (wt[i][0] - t0) / 100.0;
            hist[(int)st0 < 63 ? (int)st0 : 63]++;
            dur += (wt[i][1] - wt[i][0]) / 100.0;
            late = st0 > late ? st0 : late;
            const unsigned xcc = wt[i][2] & 0xf, hw = (unsigned)wt[i][3];
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            percu[xcc * 64 + se * 16 + sh * 8 + (cu & 7)]++;  // (layout of HW_ID is approximate: used only to count distinct slots)
        }
        {   // per XCD: mean / max workgroup duration, and first vs second workgroup of a (xcc, hw id) slot
            double sx[16] = {0}, mxx[16] = {0};
            int nx[16] = {0};
            for (int i = 0; i < nwg && i < 1024; ++i) {
                const int x = (int)(wt[i][2] & 0xf);
                const double d = (wt[i][1] - wt[i][0]) / 100.0;
                sx[x] += d, nx[x]++, mxx[x] = d > mxx[x] ? d : mxx[x];
            }
            printf("  per XCD mean (max) workgroup duration, us:");
            for (int x = 0; x < 16; ++x)
                if (nx[x]) printf(" [%d] %.1f (%.1f)", x, sx[x] / nx[x], mxx[x]);
            printf("\n");
            double lo = 1e9, hi = 0;
            int ilo = 0, ihi = 0;
            for (int i = 0; i < nwg && i < 1024; ++i) {
                const double d = (wt[i][1] - wt[i][0]) / 100.0;
                if (d < lo) lo = d, ilo = i;
                if (d > hi) hi = d, ihi = i;
            }
            printf("  fastest workgroup %d: %.1f us, slowest %d: %.1f us\n", ilo, lo, ihi, hi);
        }
        int cus = 0, mx = 0;
        for (int v : percu) { cus += v > 0; mx = v > mx ? v : mx; }
        printf("  launch span %.2f us over %d workgroups | mean workgroup duration %.2f us | last start +%.2f us | distinct (xcc, hw id) slots %d, max workgroups on one %d | starts per us:", (t1 - t0) / 100.0, nwg, dur / nwg, late, cus, mx);
        for (int i = 0; i < 16; ++i) printf(" %d", hist[i]);
        printf("\n");
    }
    return 0;
}
