// bf3bench6.hip — the round-6 bf16x3 kernel (edmp_amd/csrc/bf3.hip, six exact partial products, two accumulators per tile) against the
// production fp32-MFMA kernel wide_conv_kernel<KIND, FMS, 32, GS, LIN, RES> on the same data: error of both against a float64
// evaluation of the whole op (Conv1d k5 + bias -> GroupNorm(8) -> Mish -> + time bias, with RES also the folded residual 1x1 conv;
// KIND 1 / 2: the strided / transposed resampling conv + bias) and microseconds per launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=12 -DKINDV=0 -DLV=7 -DGSV=32 -DFMS=16 -DBMS=32 [-DRESV=1] [-DEDMP_BF3_STAMPS] tools/bf3bench6.hip -o tools/bf3bench6_x
//   tools/bf3bench6_x [Cin = Cout] [weight family 0..3] [1: the input is two concatenated halves] [B = 1024]
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
#include "../edmp_amd/csrc/bf3.hip"
using namespace edmp;
#ifndef KINDV
#define KINDV 0
#endif
#ifndef LV
#define LV 7
#endif
#ifndef GSV
#define GSV 32
#endif
#ifndef FMS
#define FMS 16
#endif
#ifndef BMS
#define BMS 32
#endif
#ifndef RESV
#define RESV 0
#endif
#ifndef FCG  // output channels per workgroup of the fp32 instance (64 at 512 channels)
#define FCG 32
#endif
#ifndef BCG  // ... of the bf16x3 instance (64 for a Conv1dBlock at 512 channels: whole GroupNorm groups)
#define BCG 32
#endif
#ifndef FKIND  // kernel form of the fp32 instance (4 = WK_K5K4: the Conv1dBlock at L = 4 in nested Karatsuba form)
#define FKIND KINDV
#endif

// canary: pure VALU work per lane (a long dependent fma / sin chain in f32 with a little f64), one result per thread: any deviation from a
// solo run while another kernel shares the CUs is a co-residency fault, not a data race (the canary touches nothing but its own output)
__global__ __launch_bounds__(256) void canary_kernel(float* out, int iters, float seed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float a = seed + 1e-3f * (float)(i & 1023), b = 0.5f, g[7] = {0, 0, 0, 0, 0, 0, 0};
    double d = 1.0;
    for (int k = 0; k < iters; ++k) {
        a = fmaf(a, 0.999f, 0.001f * b);
        b = fmaf(b, 0.998f, 0.002f * __sinf(a));
        const float cf = a > b ? a : b;
#pragma unroll
        for (int j = 0; j < 7; ++j) g[j] = fmaf(cf, b - (float)j * 0.01f, g[j] * 0.99f);
        if ((k & 63) == 0) d = d * 0.999 + (double)a;
    }
    float r = (float)d;
#pragma unroll
    for (int j = 0; j < 7; ++j) r += g[j];
    out[i] = r;
}

template <class T>
static T* up(const std::vector<T>& h) {
    T* d;
    hipMalloc((void**)&d, h.size() * sizeof(T));
    hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}

int main(int argc, char** argv) {
    constexpr int L = LV, KIND = KINDV, GS = GSV;
    constexpr bool RES = RESV != 0;
    using CF = WideCfg<FKIND, FMS, FCG, GS, L, RES>;
    using CB = Bf3Cfg<KIND, BMS, BCG, GS, L, RES>;
    constexpr int LOUT = CB::LOUT, NTAP = CB::NTAP;
    static_assert(CF::LOUT == LOUT, "same op");
    const int B = argc > 4 ? atoi(argv[4]) : 1024, C = 8 * GS, Cin = argc > 1 ? atoi(argv[1]) : C;
    const int wkind = argc > 2 ? atoi(argv[2]) : 0;  // 0 uniform taps, 1 centre tap x30, 2 heavy-tailed (Student t, 2 d.o.f.), 3 per-channel scale spread 1e-3..1e3
    const int split = argc > 3 ? atoi(argv[3]) : 0;  // 1: the input is two tensors of Cin / 2 channels each (a concatenation)
    std::mt19937 g(1);
    std::uniform_real_distribution<float> d(-1.f, 1.f);
    std::normal_distribution<float> nd(0.f, 1.f);
    const float ws = 1.7f / std::sqrt((float)NTAP * Cin);
    std::vector<float> hx((size_t)B * L * Cin), hW((size_t)6 * C * Cin, 0.f), hb(C), hg(C), hbe(C), htb(C), hrb(C);
    for (auto& v : hx) v = 1.5f * nd(g);
    for (size_t i = 0; i < (size_t)NTAP * C * Cin; ++i) hW[i] = ws * d(g);
    const size_t n = (size_t)C * Cin;
    if (wkind == 1) for (size_t i = 0; i < n; ++i) hW[(NTAP / 2) * n + i] *= 30.f;
    if (wkind == 2) for (size_t i = 0; i < NTAP * n; ++i) { const float a = nd(g), b1 = nd(g), b2 = nd(g); hW[i] = 0.3f * ws * a / std::sqrt(0.5f * (b1 * b1 + b2 * b2) + 1e-12f); }
    if (wkind == 3) for (int co = 0; co < C; ++co) { const float sc = std::pow(10.f, 3.f * d(g)); for (int t = 0; t < NTAP; ++t) for (int ci = 0; ci < Cin; ++ci) hW[((size_t)t * C + co) * Cin + ci] *= sc; }
    if (RES) for (size_t i = 0; i < n; ++i) hW[5 * n + i] = 2.f * ws * d(g);
    for (auto& v : hb) v = 0.1f * d(g);
    for (auto& v : hrb) v = 0.1f * d(g);
    for (auto& v : hg) v = 1.f + 0.5f * d(g);
    for (auto& v : hbe) v = 0.3f * d(g);
    for (auto& v : htb) v = 0.5f * d(g);
    std::vector<float> hWf((size_t)(C / FMS) * (Cin / CF::KG) * CF::NSLAB * 256);
    if (FKIND == WK_K5K4) pack_fragments_k4(hW.data(), C, Cin, RES, hWf.data());
    else pack_fragments(hW.data(), C, Cin, 0, NTAP, RES, hWf.data(), FMS);
    std::vector<unsigned short> hWb(bf3_stream_elems(C, Cin, CB::NSLOT));
    pack_fragments_bf3(hW.data(), C, Cin, NTAP, RES, hWb.data());
    // device input: one tensor, or two halves [B][L][Cin/2]
    const int C1 = split ? Cin / 2 : Cin, C2 = split ? Cin / 2 : 0;
    std::vector<float> h1((size_t)B * L * C1), h2((size_t)B * L * (C2 ? C2 : 1));
    for (size_t r = 0; r < (size_t)B * L; ++r)
        for (int c = 0; c < Cin; ++c) (c < C1 ? h1[r * C1 + c] : h2[r * C2 + c - C1]) = hx[r * Cin + c];
    float *x1 = up(h1), *x2 = up(h2), *Wf = up(hWf), *bias = up(hb), *gam = up(hg), *bet = up(hbe), *tb = up(htb), *rb = up(hrb), *y32, *y16, *r32, *r16;
    unsigned short* Wb = up(hWb);
    const size_t nout = (size_t)B * LOUT * C, nres = (size_t)B * L * C;
    hipMalloc((void**)&y32, nout * 4);
    hipMalloc((void**)&y16, nout * 4);
    hipMalloc((void**)&r32, nres * 4);
    hipMalloc((void**)&r16, nres * 4);
    hipMemset(y16, 0xff, nout * 4);
    RcbP p{};
    p.src1 = x1, p.src2 = C2 ? x2 : nullptr, p.C1 = C1, p.C2 = C2, p.W = Wf, p.bias = bias, p.gamma = gam, p.beta = bet, p.add_tb = tb, p.dst = y32, p.Cout = C, p.B = B;
    if (RES) p.res_out = r32, p.res_bias = rb;
    RcbP q = p;
    q.dst = y16;
    q.W = reinterpret_cast<const float*>(Wb);
    if (RES) q.res_out = r16;
    launch_wide_t<FKIND, FMS, FCG, GS, L, RES>(p, 0);
    launch_bf3_t<KIND, BMS, BCG, GS, L, RES>(q, 0);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
    std::vector<float> h32(nout), h16(nout), hr32(nres), hr16(nres);
    hipMemcpy(h32.data(), y32, nout * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h16.data(), y16, nout * 4, hipMemcpyDeviceToHost);
    if (RES) { hipMemcpy(hr32.data(), r32, nres * 4, hipMemcpyDeviceToHost); hipMemcpy(hr16.data(), r16, nres * 4, hipMemcpyDeviceToHost); }
    const int NS = 6;
    double e32 = 0, e16 = 0, m32 = 0, m16 = 0, dd = 0, rr = 0, er32 = 0, er16 = 0, mr32 = 0, mr16 = 0;
    for (int sidx = 0; sidx < NS; ++sidx) {
        const int b = (sidx == NS - 1) ? B - 1 : (sidx * 171) % B;
        std::vector<double> y((size_t)LOUT * C);
        for (int l = 0; l < LOUT; ++l)
            for (int co = 0; co < C; ++co) {
                double a = hb[co];
                for (int lp = 0; lp < L; ++lp) {
                    const int t = CB::slot(l, lp);
                    if (t < 0) continue;
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
                    const double a1 = hr32[o] - r, a2 = hr16[o] - r;
                    er32 += a1 * a1, er16 += a2 * a2;
                    mr32 = std::max(mr32, std::fabs(a1)), mr16 = std::max(mr16, std::fabs(a2));
                }
            }
        for (int gi = 0; gi < 8; ++gi) {
            double mu = 0, var = 0;
            if (CB::GN) {
                for (int l = 0; l < LOUT; ++l) for (int c = 0; c < GS; ++c) mu += y[(size_t)l * C + gi * GS + c];
                mu /= (double)(LOUT * GS);
                for (int l = 0; l < LOUT; ++l) for (int c = 0; c < GS; ++c) { const double t = y[(size_t)l * C + gi * GS + c] - mu; var += t * t; }
                var /= (double)(LOUT * GS);
            }
            for (int l = 0; l < LOUT; ++l)
                for (int c = 0; c < GS; ++c) {
                    const int ch = gi * GS + c;
                    double ref = y[(size_t)l * C + ch];
                    if (CB::GN) {
                        const double z = (ref - mu) / std::sqrt(var + 1e-5) * hg[ch] + hbe[ch];
                        ref = z * std::tanh(std::log1p(std::exp(z))) + htb[ch];
                    }
                    const size_t o = ((size_t)b * LOUT + l) * C + ch;
                    const double a = h32[o] - ref, c2 = h16[o] - ref;
                    e32 += a * a, e16 += c2 * c2, rr += ref * ref;
                    m32 = std::max(m32, std::fabs(a)), m16 = std::max(m16, std::fabs(c2));
                }
        }
    }
    size_t nbad = 0;
    for (size_t i = 0; i < nout; ++i) {
        const double df = std::fabs((double)h32[i] - (double)h16[i]);
        if (!(df <= 1e-3)) ++nbad;
        if (df == df) dd = std::max(dd, df);
    }
    const double cnt = (double)NS * LOUT * C;
    printf("KIND %d L %d GS %d RES %d Cin %d%s weights %d B %d | fp32 MS %d, bf16x3 MS %d (part 0 = positions 0x%x, %d + %d tiles) | rms(out) %.3f | vs float64 (6 samples): fp32-MFMA rmse %.3e max %.3e | bf16x3 rmse %.3e max %.3e (x%.2f / x%.2f) | max |fp32 - bf16x3| over all %.3e, %zu elements off by > 1e-3",
           KIND, L, GS, (int)RES, Cin, split ? " (two halves)" : "", wkind, B, FMS, BMS, CB::P0, CB::ntiles(0), CB::ntiles(1), std::sqrt(rr / cnt), std::sqrt(e32 / cnt), m32, std::sqrt(e16 / cnt), m16, std::sqrt(e16 / e32), m16 / m32, dd, nbad);
    if (RES) printf(" | residual conv rmse / max: fp32 %.3e / %.3e bf16x3 %.3e / %.3e", std::sqrt(er32 / cnt), mr32, std::sqrt(er16 / cnt), mr16);
    printf("\n");
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best32 = 1e9f, best16 = 1e9f;
    for (int r = 0; r < 6; ++r) {
        float ms;
        hipEventRecord(e0, 0);
        for (int i = 0; i < 50; ++i) launch_wide_t<FKIND, FMS, FCG, GS, L, RES>(p, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        best32 = std::min(best32, ms * 20.f);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 50; ++i) launch_bf3_t<KIND, BMS, BCG, GS, L, RES>(q, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        best16 = std::min(best16, ms * 20.f);
    }
    printf("  us per launch (chains of 50): fp32-MFMA %.2f | bf16x3 %.2f | x%.3f\n", best32, best16, best32 / best16);
    {   // co-residency check: the same launch on 4 streams at once (quarter batches, as the sampler's row chains do), many rounds, every output bit-compared with the serial one
        hipStream_t st[4];
        float* yo[4];
        for (int k = 0; k < 4; ++k) { hipStreamCreate(&st[k]); hipMalloc((void**)&yo[k], nout * 4); }
        size_t bad = 0;
        std::vector<float> hk(nout);
        const int Bq = B / 4;
        for (int round = 0; round < 40; ++round) {
            for (int k = 0; k < 4; ++k) {
                RcbP qq = q;
                qq.B = Bq;
                qq.src1 = q.src1 + (size_t)k * Bq * L * C1;
                if (C2) qq.src2 = q.src2 + (size_t)k * Bq * L * C2;
                qq.dst = yo[k] + (size_t)k * Bq * LOUT * C;
                if (RES) qq.res_out = r16 + (size_t)k * Bq * L * C;
                for (int rep = 0; rep < 3; ++rep) launch_bf3_t<KIND, BMS, BCG, GS, L, RES>(qq, st[k]);
            }
            hipDeviceSynchronize();
            for (int k = 0; k < 4; ++k) {
                hipMemcpy(hk.data(), yo[k], nout * 4, hipMemcpyDeviceToHost);
                for (size_t i = (size_t)k * Bq * LOUT * C; i < (size_t)(k + 1) * Bq * LOUT * C; ++i) bad += (__builtin_bit_cast(unsigned, hk[i]) != __builtin_bit_cast(unsigned, h16[i]));
            }
        }
        printf("  4 quarter-batch launches on 4 streams x 40 rounds: %zu elements differ from the serial launch\n", bad);
        // mixed: the bf16x3 kernel on two streams beside the fp32 kernel on two others
        size_t badb = 0, badf = 0;
        for (int round = 0; round < 40; ++round) {
            for (int k = 0; k < 4; ++k) {
                RcbP qq = (k & 1) ? q : p;
                qq.B = Bq;
                qq.src1 = q.src1 + (size_t)k * Bq * L * C1;
                if (C2) qq.src2 = q.src2 + (size_t)k * Bq * L * C2;
                qq.dst = yo[k] + (size_t)k * Bq * LOUT * C;
                if (RES) qq.res_out = ((k & 1) ? r16 : r32) + (size_t)k * Bq * L * C;
                for (int rep = 0; rep < 3; ++rep) {
                    if (k & 1) launch_bf3_t<KIND, BMS, BCG, GS, L, RES>(qq, st[k]);
                    else launch_wide_t<FKIND, FMS, FCG, GS, L, RES>(qq, st[k]);
                }
            }
            hipDeviceSynchronize();
            for (int k = 0; k < 4; ++k) {
                hipMemcpy(hk.data(), yo[k], nout * 4, hipMemcpyDeviceToHost);
                const std::vector<float>& ref = (k & 1) ? h16 : h32;
                for (size_t i = (size_t)k * Bq * LOUT * C; i < (size_t)(k + 1) * Bq * LOUT * C; ++i) ((k & 1) ? badb : badf) += (__builtin_bit_cast(unsigned, hk[i]) != __builtin_bit_cast(unsigned, ref[i]));
            }
        }
        printf("  mixed (bf16x3 on 2 streams + fp32 on 2 streams) x 40 rounds: bf16x3 outputs differ in %zu elements, fp32 outputs in %zu\n", badb, badf);
        // canary beside each kernel
        const int NC = 1024 * 256, IT = 6000;
        float* cd;
        hipMalloc((void**)&cd, NC * 4);
        std::vector<float> c0(NC), c1(NC);
        hipLaunchKernelGGL(canary_kernel, dim3(NC / 256), dim3(256), 0, st[0], cd, IT, 0.3f);
        hipDeviceSynchronize();
        hipMemcpy(c0.data(), cd, NC * 4, hipMemcpyDeviceToHost);
        for (int which = 0; which < 2; ++which) {
            size_t badc = 0;
            int lanes[64] = {0};
            for (int round = 0; round < 60; ++round) {
                hipMemsetAsync(cd, 0, NC * 4, st[0]);
                hipLaunchKernelGGL(canary_kernel, dim3(NC / 256), dim3(256), 0, st[0], cd, IT, 0.3f);
                for (int rep = 0; rep < 12; ++rep) {
                    if (which == 0) launch_bf3_t<KIND, BMS, BCG, GS, L, RES>(q, st[1]);
                    else launch_wide_t<FKIND, FMS, FCG, GS, L, RES>(p, st[1]);
                }
                hipDeviceSynchronize();
                hipMemcpy(c1.data(), cd, NC * 4, hipMemcpyDeviceToHost);
                for (int i = 0; i < NC; ++i)
                    if (__builtin_bit_cast(unsigned, c1[i]) != __builtin_bit_cast(unsigned, c0[i])) { ++badc; ++lanes[i & 63]; }
            }
            printf("  VALU canary beside the %s kernel x 60 rounds: %zu of %d x 60 results differ from the solo run; lanes hit:", which == 0 ? "bf16x3" : "fp32-MFMA", badc, NC);
            for (int l = 0; l < 64; ++l) if (lanes[l]) printf(" %d(x%d)", l, lanes[l]);
            printf("\n");
        }
    }
#ifdef EDMP_BF3_STAMPS
    long long st[8][8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(edmp::g_bf3_stamps), sizeof(st));
    for (int w = 0; w < 8; ++w)
        printf("  wave %d (%s): prologue %lld | K loop %lld | spill %lld | final pass %lld cycles\n", w, w < 4 ? "mfma" : "stage", st[w][1] - st[w][0], st[w][2] - st[w][1], st[w][3] - st[w][2], st[w][4] - st[w][3]);
#endif
    return nbad ? 2 : 0;
}
