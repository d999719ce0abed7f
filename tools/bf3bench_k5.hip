// bf3bench_k5.hip — the bf16x3 exact-product experiment on the direct-form Conv1dBlock of the 256-channel levels (tools/wide_bf3_k5.hip)
// against the production fp32-MFMA kernel wide_conv_kernel<WK_K5 | WK_K5K4, 32, 32, 32, LIN, RES> on the same data: error of both
// against a float64 evaluation of the whole op (Conv1d k5 + bias -> GroupNorm(8) -> Mish -> + time bias; blocks.py:13-34; with RES
// also the folded residual 1x1 conv, blocks.py:147-152) and microseconds per launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=12 -DLV=7 [-DRESV=1] [-DBF3_STAMPS] tools/bf3bench_k5.hip -o tools/bf3bench_k5
//   tools/bf3bench_k5 [Cin = 256] [weight family 0..3]
#include "../edmp_amd/csrc/common.h"
#include "../edmp_amd/csrc/params.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
namespace edmp {
void set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fprintf(stderr, "\n"); }
}
#include "../edmp_amd/csrc/wide.hip"
#include "wide_bf3_k5.hip"
using namespace edmp;
#ifndef LV
#define LV 7
#endif
#ifndef RESV
#define RESV 0
#endif
#ifndef FP32KIND  // the production kernel this instance runs on: WK_K5 (L = 7) or WK_K5K4 (L = 4, nested Karatsuba form)
#define FP32KIND (LV == 4 ? WK_K5K4 : WK_K5)
#endif

template <class T>
static T* up(const std::vector<T>& h) {
    T* d;
    hipMalloc((void**)&d, h.size() * sizeof(T));
    hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}

int main(int argc, char** argv) {
    constexpr int L = LV;
    constexpr bool RES = RESV != 0;
    const int B = 1024, C = 256, Cin = argc > 1 ? atoi(argv[1]) : 256;
    const int wkind = argc > 2 ? atoi(argv[2]) : 0;  // 0 uniform taps, 1 centre tap x30, 2 heavy-tailed (Student t, 2 d.o.f.), 3 per-channel scale spread 1e-3..1e3
    std::mt19937 g(1);
    std::uniform_real_distribution<float> d(-1.f, 1.f);
    std::normal_distribution<float> nd(0.f, 1.f);
    const float ws = 1.7f / std::sqrt(5.0f * Cin);
    std::vector<float> hx((size_t)B * L * Cin), hW((size_t)6 * C * Cin, 0.f), hb(C), hg(C), hbe(C), htb(C), hrb(C);
    for (auto& v : hx) v = 1.5f * nd(g);
    for (size_t i = 0; i < (size_t)5 * C * Cin; ++i) hW[i] = ws * d(g);
    const size_t n = (size_t)C * Cin;
    if (wkind == 1) for (size_t i = 0; i < n; ++i) hW[2 * n + i] *= 30.f;
    if (wkind == 2) for (size_t i = 0; i < 5 * n; ++i) { const float a = nd(g), b1 = nd(g), b2 = nd(g); hW[i] = 0.3f * ws * a / std::sqrt(0.5f * (b1 * b1 + b2 * b2) + 1e-12f); }
    if (wkind == 3) for (int co = 0; co < C; ++co) { const float sc = std::pow(10.f, 3.f * d(g)); for (int t = 0; t < 5; ++t) for (int ci = 0; ci < Cin; ++ci) hW[((size_t)t * C + co) * Cin + ci] *= sc; }
    if (RES) for (size_t i = 0; i < n; ++i) hW[5 * n + i] = 2.f * ws * d(g);
    for (auto& v : hb) v = 0.1f * d(g);
    for (auto& v : hrb) v = 0.1f * d(g);
    for (auto& v : hg) v = 1.f + 0.5f * d(g);
    for (auto& v : hbe) v = 0.3f * d(g);
    for (auto& v : htb) v = 0.5f * d(g);
    using C32 = WideCfg<FP32KIND, 32, 32, 32, L, RES>;
    std::vector<float> hWf((size_t)(C / 32) * (Cin / 8) * C32::NSLAB * 256);
    if (FP32KIND == WK_K5K4) pack_fragments_k4(hW.data(), C, Cin, RES, hWf.data());
    else pack_fragments(hW.data(), C, Cin, 0, 5, RES, hWf.data(), 32);
    using CB = K5Bf3Cfg<L, RES>;
    std::vector<unsigned short> hWb((size_t)(C / 32) * (Cin / 16) * CB::NSLOT * 3 * 512);
    pack_fragments_k5_bf3(hW.data(), C, Cin, RES, hWb.data());
    float *x = up(hx), *Wf = up(hWf), *bias = up(hb), *gam = up(hg), *bet = up(hbe), *tb = up(htb), *rb = up(hrb), *y32, *y16, *r32, *r16;
    unsigned short* Wb = up(hWb);
    const size_t nout = (size_t)B * L * C;
    hipMalloc((void**)&y32, nout * 4);
    hipMalloc((void**)&y16, nout * 4);
    hipMalloc((void**)&r32, nout * 4);
    hipMalloc((void**)&r16, nout * 4);
    RcbP p{};
    p.src1 = x, p.C1 = Cin, p.W = Wf, p.bias = bias, p.gamma = gam, p.beta = bet, p.add_tb = tb, p.dst = y32, p.Cout = C, p.B = B;
    if (RES) p.res_out = r32, p.res_bias = rb;
    RcbP q = p;
    q.dst = y16;
    if (RES) q.res_out = r16;
    launch_wide_t<FP32KIND, 32, 32, 32, L, RES>(p, 0);
    launch_k5_bf3<L, RES>(q, Wb, 0);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
    std::vector<float> h32(nout), h16(nout), hr32(nout), hr16(nout);
    hipMemcpy(h32.data(), y32, nout * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h16.data(), y16, nout * 4, hipMemcpyDeviceToHost);
    if (RES) { hipMemcpy(hr32.data(), r32, nout * 4, hipMemcpyDeviceToHost); hipMemcpy(hr16.data(), r16, nout * 4, hipMemcpyDeviceToHost); }
    const int NS = 6;
    double e32 = 0, e16 = 0, m32 = 0, m16 = 0, dd = 0, rr = 0, er32 = 0, er16 = 0;
    for (int sidx = 0; sidx < NS; ++sidx) {
        const int b = sidx * 171 % B;
        std::vector<double> y((size_t)L * C);
        for (int l = 0; l < L; ++l)
            for (int co = 0; co < C; ++co) {
                double a = hb[co];
                for (int t = 0; t < 5; ++t) {
                    const int lp = l + t - 2;
                    if (lp < 0 || lp >= L) continue;
                    const float* w = &hW[((size_t)t * C + co) * Cin];
                    const float* xv = &hx[((size_t)b * L + lp) * Cin];
                    for (int ci = 0; ci < Cin; ++ci) a += (double)w[ci] * (double)xv[ci];
                }
                y[(size_t)l * C + co] = a;
                if (RES) {
                    double r = hrb[co];
                    const float* w = &hW[((size_t)5 * C + co) * Cin];
                    const float* xv = &hx[((size_t)b * L + l) * Cin];
                    for (int ci = 0; ci < Cin; ++ci) r += (double)w[ci] * (double)xv[ci];
                    const size_t o = ((size_t)b * L + l) * C + co;
                    er32 = std::max(er32, std::fabs(hr32[o] - r)), er16 = std::max(er16, std::fabs(hr16[o] - r));
                }
            }
        for (int gi = 0; gi < 8; ++gi) {
            double mu = 0, var = 0;
            for (int l = 0; l < L; ++l) for (int c = 0; c < 32; ++c) mu += y[(size_t)l * C + gi * 32 + c];
            mu /= (double)(L * 32);
            for (int l = 0; l < L; ++l) for (int c = 0; c < 32; ++c) { const double t = y[(size_t)l * C + gi * 32 + c] - mu; var += t * t; }
            var /= (double)(L * 32);
            for (int l = 0; l < L; ++l)
                for (int c = 0; c < 32; ++c) {
                    const int ch = gi * 32 + c;
                    const double z = (y[(size_t)l * C + ch] - mu) / std::sqrt(var + 1e-5) * hg[ch] + hbe[ch];
                    const double ref = z * std::tanh(std::log1p(std::exp(z))) + htb[ch];
                    const size_t o = ((size_t)b * L + l) * C + ch;
                    const double a = h32[o] - ref, c2 = h16[o] - ref;
                    e32 += a * a, e16 += c2 * c2, rr += ref * ref;
                    m32 = std::max(m32, std::fabs(a)), m16 = std::max(m16, std::fabs(c2));
                }
        }
    }
    for (size_t i = 0; i < nout; ++i) dd = std::max(dd, (double)std::fabs(h32[i] - h16[i]));
    const double cnt = (double)NS * L * C;
    printf("L %d RES %d Cin %d weights %d | rms(out) %.3f | vs float64 (6 samples): fp32-MFMA rmse %.3e max %.3e | bf16x3 rmse %.3e max %.3e | max |fp32 - bf16x3| over all %.3e", L, (int)RES, Cin, wkind,
           std::sqrt(rr / cnt), std::sqrt(e32 / cnt), m32, std::sqrt(e16 / cnt), m16, dd);
    if (RES) printf(" | residual conv max err fp32 %.3e bf16x3 %.3e", er32, er16);
    printf("\n");
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best32 = 1e9f, best16 = 1e9f;
    for (int r = 0; r < 6; ++r) {
        float ms;
        hipEventRecord(e0, 0);
        for (int i = 0; i < 50; ++i) launch_wide_t<FP32KIND, 32, 32, 32, L, RES>(p, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        best32 = std::min(best32, ms * 20.f);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 50; ++i) launch_k5_bf3<L, RES>(q, Wb, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        best16 = std::min(best16, ms * 20.f);
    }
    printf("us per launch (chains of 50): fp32-MFMA %.2f | bf16x3 %.2f | x%.3f\n", best32, best16, best32 / best16);
#ifdef BF3_STAMPS
    {   // one launch alone: span from the first workgroup's start to the last one's end
        unsigned long long z[2] = {~0ull, 0ull}, w2[2];
        hipMemcpyToSymbol(HIP_SYMBOL(edmp::g_bf3_wall), z, sizeof(z));
        for (int i = 0; i < 49; ++i) launch_k5_bf3<L, RES>(q, Wb, 0);  // warm chain: the stamps below are the 50th launch's
        hipDeviceSynchronize();
        hipMemcpyToSymbol(HIP_SYMBOL(edmp::g_bf3_wall), z, sizeof(z));
        for (int i = 0; i < 50; ++i) launch_k5_bf3<L, RES>(q, Wb, 0);
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(w2, HIP_SYMBOL(edmp::g_bf3_wall), sizeof(w2));
        printf("  bf16x3 chain of 50: first workgroup start -> last workgroup end %.2f us per launch\n", (double)(w2[1] - w2[0]) / 100.0 / 50.0);
    }
    long long st[8][8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(edmp::g_bf3_stamps), sizeof(st));
    for (int w = 0; w < 8; ++w)
        printf("  wave %d (%s): prologue %lld | K loop %lld | epilogue %lld cycles | whole kernel %.2f us wall = %.2f GHz\n", w, w < 4 ? "mfma" : "stage", st[w][1] - st[w][0], st[w][2] - st[w][1],
               st[w][3] - st[w][2], (st[w][7] - st[w][6]) / 100.0, (double)(st[w][3] - st[w][0]) / ((st[w][7] - st[w][6]) * 10.0));
#endif
    return 0;
}
