// kbench.hip — standalone A/B harness for the UNet conv kernels (no Python, no torch): old vs new kernel of every wide
// shape of the full-size net at B = 1024 on the same random data; prints max |diff|, microseconds per launch
// (interleaved rounds, HIP events) and executed TFLOP/s.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kbench.hip -o tools/kbench
#include "../edmp_amd/csrc/libedmp_hip.hip"
#include "legacy_rcb_conv_r1.hip"

#include <algorithm>
#include <cmath>
#include <random>

using namespace edmp;

static std::vector<float> rnd(size_t n, uint32_t seed, float scale) {
    std::mt19937 g(seed);
    std::uniform_real_distribution<float> d(-scale, scale);
    std::vector<float> v(n);
    for (auto& x : v) x = d(g);
    return v;
}
template <class T>
static T* up(const std::vector<T>& h) {
    T* d = nullptr;
    hipMalloc((void**)&d, h.size() * sizeof(T));
    hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}
static double max_abs_diff(const float* a_dev, const float* b_dev, size_t n, double* ref_max) {
    std::vector<float> a(n), b(n);
    hipMemcpy(a.data(), a_dev, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), b_dev, n * 4, hipMemcpyDeviceToHost);
    double m = 0, r = 0;
    for (size_t i = 0; i < n; ++i) {
        m = std::max(m, (double)std::fabs(a[i] - b[i]));
        r = std::max(r, (double)std::fabs(a[i]));
        if (std::isnan(a[i]) || std::isnan(b[i])) m = 1e30;
    }
    *ref_max = r;
    return m;
}

template <class F>
static float time_us(F&& f, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ms * 1000.f / iters;
}

// Conv1dBlock (k5 + GroupNorm + Mish + add [+ folded residual 1x1 conv]): round-1 kernel(s) vs the position-tile kernel.
// MS = 32: reference = legacy rcb_conv_kernel; MS = 16 (128-channel levels): reference = rcb_rows_kernel (+ conv_mfma for the
// residual 1x1 conv, which round 1 ran as its own launch)
template <int MS, int CG, int GS, int L, bool RES, int KIND = WK_K5>
static void run_wide(int B, int C1, int C2, bool with_res_add) {
    const int Cout = GS * 8, Cin = C1 + C2;
    const int ntap_store = 6;
    const float wscale = 1.0f / std::sqrt((float)Cin * 5);
    auto hW = rnd((size_t)ntap_store * Cout * Cin, 1, wscale * 1.7f);
    auto hx1 = rnd((size_t)B * L * C1, 2, 1.5f);
    auto hx2 = rnd((size_t)B * L * std::max(C2, 1), 3, 1.5f);
    auto hb = rnd(Cout, 4, 0.1f), hg = rnd(Cout, 5, 1.0f), hbe = rnd(Cout, 6, 0.3f), htb = rnd(Cout, 7, 0.5f), hrb = rnd(Cout, 8, 0.1f);
    auto hres = rnd((size_t)B * L * Cout, 9, 1.0f);
    using Cf = WideCfg<KIND, MS, CG, GS, L, RES>;
    std::vector<float> hWf((size_t)(Cout / Cf::SW) * (Cin / Cf::KG) * Cf::NSLAB * 256);
    if constexpr (KIND == WK_K5K2) pack_fragments_k2(hW.data(), Cout, Cin, RES, hWf.data());
    else if constexpr (KIND == WK_K5K4) pack_fragments_k4(hW.data(), Cout, Cin, RES, hWf.data());
    else pack_fragments(hW.data(), Cout, Cin, Cf::KT0, Cf::NTAP, RES, hWf.data(), Cf::SW);
    float *W = up(hW), *Wf = up(hWf), *x1 = up(hx1), *x2 = up(hx2), *bias = up(hb), *gam = up(hg), *bet = up(hbe), *tb = up(htb), *rb = up(hrb), *res = up(hres);
    float *d_old, *d_new, *r_old, *r_new;
    const size_t nout = (size_t)B * L * Cout;
    hipMalloc((void**)&d_old, nout * 4);
    hipMalloc((void**)&d_new, nout * 4);
    hipMalloc((void**)&r_old, nout * 4);
    hipMalloc((void**)&r_new, nout * 4);
    hipMemset(d_old, 0, nout * 4);
    hipMemset(d_new, 0xff, nout * 4);
    RcbP p{};
    p.src1 = x1;
    p.src2 = C2 ? x2 : nullptr;
    p.C1 = C1;
    p.C2 = C2;
    p.W = W;
    p.bias = bias;
    p.gamma = gam;
    p.beta = bet;
    p.add_tb = with_res_add ? nullptr : tb;
    p.add_res = with_res_add ? res : nullptr;
    p.dst = d_old;
    p.Cout = Cout;
    p.B = B;
    p.res_out = (RES && MS == 32) ? r_old : nullptr;
    p.res_bias = RES ? rb : nullptr;
    RcbP pn = p;
    pn.W = Wf;
    pn.dst = d_new;
    pn.res_out = RES ? r_new : nullptr;
    ConvP c{};  // the residual 1x1 conv as round 1 ran it on the 128-channel levels
    c.src1 = x1;
    c.src2 = C2 ? x2 : nullptr;
    c.C1 = C1;
    c.C2 = C2;
    c.Lin = L;
    c.Lout = L;
    c.ntaps = 1;
    c.stride = 1;
    c.pad = 0;
    c.transposed = 0;
    c.W = W + (size_t)5 * Cout * Cin;
    c.bias = rb;
    c.dst = r_old;
    c.Cout = Cout;
    c.B = B;
    auto f_old = [&] {
        if constexpr (MS == 32) launch_rcb_t<CG, L, RES>(p, 0);
        else {
            launch_rows(p, rows_variant(Cout, L, C1, C2), 0);
            if (RES) launch_conv(c, 0);
        }
    };
    auto f_new = [&] { launch_wide_t<KIND, MS, CG, GS, L, RES>(pn, 0); };
    f_old();
    f_new();
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        printf("<%d,%d,%d,%d> C1=%d C2=%d: launch failed: %s\n", MS, GS, L, (int)RES, C1, C2, hipGetErrorString(e));
        return;
    }
    double rm = 0, rm2 = 0;
    const double d1 = max_abs_diff(d_old, d_new, nout, &rm);
    const double d2 = RES ? max_abs_diff(r_old, r_new, nout, &rm2) : 0.0;
    float t_old = 1e9f, t_new = 1e9f;
    for (int round = 0; round < 5; ++round) {
        t_old = std::min(t_old, time_us(f_old, 50));
        t_new = std::min(t_new, time_us(f_new, 50));
    }
    const double fl = 2.0 * B * (double)Cout * Cin * (Cf::valid_pairs() + (RES ? L : 0));
    printf("k5%s<MS%d,cg%2d,L%2d,res%d> Cin=%4d+%4d  max|d| %.2e (ref %.1f) res %.2e | old %7.2f us %6.1f TF | new %7.2f us %6.1f TF (%.3f of 157.3) | x%.3f\n", KIND == WK_K5K2 ? "-karatsuba" : KIND == WK_K5K4 ? "-karatsuba4" : "", MS, GS, L,
           (int)RES, C1, C2, d1, rm, d2, t_old, fl / t_old / 1e6, t_new, fl / t_new / 1e6, fl / t_new / 1e6 / 157.3, t_old / t_new);
#ifdef EDMP_STAMPS
    {
        for (int i = 0; i < 20; ++i) f_new();
        hipDeviceSynchronize();
        unsigned long long st[8][16];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(edmp::g_stamps), sizeof(st));
        printf("      stamps (shader cycles): prologue %llu | K loop %llu | spill %llu | stats+store %llu | total %llu cyc = %.2f us\n", st[0][2] - st[0][0],
               st[0][4] - st[0][2], st[0][6] - st[0][4], st[0][8] - st[0][6], st[0][8] - st[0][0], (st[0][9] - st[0][1]) / 100.0);
        printf("      K loop per wave (cycles): %llu %llu %llu %llu | of which at the chunk barrier: %llu %llu %llu %llu\n", st[5][4], st[5][5], st[5][6], st[5][7], st[5][0], st[5][1], st[5][2], st[5][3]);
    }
#endif
    for (float* q : {W, Wf, x1, x2, bias, gam, bet, tb, rb, res, d_old, d_new, r_old, r_new}) hipFree(q);
}

// down/up-sampling conv of a wide level: generic implicit-GEMM kernel (round 1) vs the position-tile kernel
template <int KIND, int MS, int CG, int GS, int LIN>
static void run_rs(int B) {
    using Cf = WideCfg<KIND, MS, CG, GS, LIN, false>;
    const int Cout = GS * 8, Cin = Cout, k = Cf::NTAP, Lout = Cf::LOUT;
    const bool tr = KIND == WK_UP;
    const float wscale = 1.0f / std::sqrt((float)Cin * k);
    auto hW = rnd((size_t)k * Cout * Cin, 11, wscale * 1.7f);  // [tap][Cout][Cin]
    auto hx = rnd((size_t)B * LIN * Cin, 12, 1.5f);
    auto hb = rnd(Cout, 13, 0.1f);
    std::vector<float> hWf((size_t)(Cout / Cf::SW) * (Cin / Cf::KG) * k * 256);
    pack_fragments(hW.data(), Cout, Cin, 0, k, false, hWf.data(), Cf::SW);
    float *W = up(hW), *Wf = up(hWf), *x = up(hx), *bias = up(hb);
    const size_t nout = (size_t)B * Lout * Cout;
    float *d_old, *d_new;
    hipMalloc((void**)&d_old, nout * 4);
    hipMalloc((void**)&d_new, nout * 4);
    hipMemset(d_new, 0xff, nout * 4);
    ConvP c{};
    c.src1 = x;
    c.C1 = Cin;
    c.Lin = LIN;
    c.Lout = Lout;
    c.ntaps = k;
    c.stride = 2;
    c.pad = 1;
    c.transposed = tr;
    c.W = W;
    c.bias = bias;
    c.dst = d_old;
    c.Cout = Cout;
    c.B = B;
    RcbP p{};
    p.src1 = x;
    p.C1 = Cin;
    p.W = Wf;
    p.bias = bias;
    p.dst = d_new;
    p.Cout = Cout;
    p.B = B;
    auto f_old = [&] { launch_conv(c, 0); };
    auto f_new = [&] { launch_wide_t<KIND, MS, CG, GS, LIN, false>(p, 0); };
    f_old();
    f_new();
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        printf("rs<%d,%d,%d>: launch failed: %s\n", KIND, GS, LIN, hipGetErrorString(e));
        return;
    }
    double rm = 0;
    const double d1 = max_abs_diff(d_old, d_new, nout, &rm);
    float t_old = 1e9f, t_new = 1e9f;
    for (int round = 0; round < 5; ++round) {
        t_old = std::min(t_old, time_us(f_old, 50));
        t_new = std::min(t_new, time_us(f_new, 50));
    }
    const double fl = 2.0 * B * (double)Cout * Cin * Cf::valid_pairs();
    printf("%s<%2d,Lin %d->%d>  C=%4d  max|d| %.2e (ref %.1f) | old %7.2f us %6.1f TF | new %7.2f us %6.1f TF (%.3f of 157.3) | x%.3f\n", tr ? "convT k4s2" : "conv  k3s2",
           CG, LIN, Lout, Cin, d1, rm, t_old, fl / t_old / 1e6, t_new, fl / t_new / 1e6, fl / t_new / 1e6 / 157.3, t_old / t_new);
    for (float* q : {W, Wf, x, bias, d_old, d_new}) hipFree(q);
}

__global__ void spin_kernel(long long cycles) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
}

// Experiment: does running two half-batch chains on two streams (16-sample workgroups, 256 per launch, two co-resident
// per CU from DIFFERENT launches) hide the per-launch fixed costs?  One chain of NCH launches of the 32-sample kernel at
// B rows vs two chains of NCH launches of the 16-sample kernel at B/2 rows each, the second stream offset by one launch.
template <int CG, int GS, int L>
static void run_dual(int B, int C1, int nch) {
    const int Cout = GS * 8, Cin = C1;
    const float wscale = 1.0f / std::sqrt((float)Cin * 5);
    auto hW = rnd((size_t)6 * Cout * Cin, 1, wscale * 1.7f);
    auto hx1 = rnd((size_t)B * L * C1, 2, 1.5f);
    auto hb = rnd(Cout, 4, 0.1f), hg = rnd(Cout, 5, 1.0f), hbe = rnd(Cout, 6, 0.3f), htb = rnd(Cout, 7, 0.5f);
    using C32 = WideCfg<WK_K5, 32, CG, GS, L, false>;
    using C16 = WideCfg<WK_K5, 16, CG, GS, L, false>;
    std::vector<float> hWf32((size_t)(Cout / 32) * (Cin / 8) * C32::NSLAB * 256), hWf16((size_t)(Cout / 16) * (Cin / 16) * C16::NSLAB * 256);
    pack_fragments(hW.data(), Cout, Cin, C32::KT0, C32::NTAP, false, hWf32.data(), 32);
    pack_fragments(hW.data(), Cout, Cin, C16::KT0, C16::NTAP, false, hWf16.data(), 16);
    float *Wf32 = up(hWf32), *Wf16 = up(hWf16), *x1 = up(hx1), *bias = up(hb), *gam = up(hg), *bet = up(hbe), *tb = up(htb);
    const size_t nout = (size_t)B * L * Cout;
    float *d32, *d16;
    hipMalloc((void**)&d32, nout * 4);
    hipMalloc((void**)&d16, nout * 4);
    RcbP p{};
    p.src1 = x1;
    p.C1 = C1;
    p.W = Wf32;
    p.bias = bias;
    p.gamma = gam;
    p.beta = bet;
    p.add_tb = tb;
    p.dst = d32;
    p.Cout = Cout;
    p.B = B;
    RcbP pa = p, pb = p;
    pa.W = pb.W = Wf16;
    pa.B = pb.B = B / 2;
    pa.dst = d16;
    pb.src1 = x1 + (size_t)(B / 2) * L * C1;
    pb.dst = d16 + (size_t)(B / 2) * L * Cout;
    hipStream_t sa, sb;
    hipStreamCreate(&sa);
    hipStreamCreate(&sb);
    hipEvent_t e0, e1, ea, eb;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventCreateWithFlags(&ea, hipEventDisableTiming);
    hipEventCreateWithFlags(&eb, hipEventDisableTiming);
    auto single = [&] {
        for (int i = 0; i < nch; ++i) launch_wide_t<WK_K5, 32, CG, GS, L, false>(p, sa);
    };
    auto single16 = [&] {  // the 16-sample kernel on the whole batch (512 workgroups per launch, one stream)
        RcbP q = pa;
        q.B = B;
        for (int i = 0; i < nch; ++i) launch_wide_t<WK_K5, 16, CG, GS, L, false>(q, sa);
    };
    auto dual = [&](int offset_us) {
        hipEventRecord(ea, sa);
        hipStreamWaitEvent(sb, ea, 0);  // fork
        if (offset_us) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, sb, (long long)offset_us * 2100);  // stream b starts late
        for (int i = 0; i < nch; ++i) launch_wide_t<WK_K5, 16, CG, GS, L, false>(pa, sa);
        for (int i = 0; i < nch; ++i) launch_wide_t<WK_K5, 16, CG, GS, L, false>(pb, sb);
        hipEventRecord(eb, sb);
        hipStreamWaitEvent(sa, eb, 0);  // join
    };
    auto timeit = [&](auto&& f) {
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0, sa);
            f();
            hipEventRecord(e1, sa);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, ms * 1000.f);
        }
        return best;
    };
    single();
    dual(5);
    hipDeviceSynchronize();
    double rm = 0;
    const double d = max_abs_diff(d32, d16, nout, &rm);
    const float t1 = timeit(single), t16 = timeit(single16), t2 = timeit([&] { dual(0); });
    printf("dual<cg%d,L%d> Cin=%d x%d launches: max|d| %.2e | 1 stream MS32 %.1f us (%.2f/launch) | 1 stream MS16 %.1f | 2 streams MS16 aligned %.1f |", GS, L, C1, nch, d, t1, t1 / nch, t16, t2);
    for (int off : {3, 6, 9, 12, 18, 25}) {
        const float t3 = timeit([&] { dual(off); });
        printf(" +%dus: %.1f (x%.3f)", off, t3, t1 / t3);
    }
    printf("\n");
    for (float* q : {Wf32, Wf16, x1, bias, gam, bet, tb, d32, d16}) hipFree(q);
    hipStreamDestroy(sa);
    hipStreamDestroy(sb);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1024;
    if (argc > 2) {
        run_dual<64, 64, 2>(B, 512, 20);
        run_dual<64, 64, 4>(B, 512, 12);
        run_dual<32, 32, 7>(B, 256, 12);
        return 0;
    }
    // 128-channel levels (16-sample tiles, 16x16x4 MFMA)
    run_wide<16, 32, 16, 13, false>(B, 128, 0, true);
    run_wide<16, 32, 16, 13, true>(B, 64, 0, false);
    run_wide<16, 32, 16, 7, false>(B, 128, 0, true);
    run_wide<16, 32, 16, 7, true>(B, 256, 256, false);
    run_rs<WK_DOWN, 16, 32, 16, 13>(B);
    run_rs<WK_UP, 16, 32, 16, 7>(B);
    // >= 256 channels (32-sample tiles, 32x32x2 MFMA)
    run_wide<32, 64, 64, 2, false>(B, 512, 0, true);
    run_wide<32, 64, 64, 2, true>(B, 512, 512, false);
    run_wide<32, 64, 64, 2, false, WK_K5K2>(B, 512, 0, true);
    run_wide<32, 64, 64, 2, true, WK_K5K2>(B, 512, 512, false);
    run_wide<32, 64, 64, 4, false, WK_K5K4>(B, 512, 0, false);
    run_wide<32, 64, 64, 4, true, WK_K5K4>(B, 256, 0, false);
    run_wide<32, 32, 32, 4, false, WK_K5K4>(B, 256, 0, true);
    run_wide<32, 32, 32, 4, true, WK_K5K4>(B, 512, 512, false);
    run_wide<32, 64, 64, 4, false>(B, 512, 0, false);
    run_wide<32, 64, 64, 4, true>(B, 256, 0, false);
    run_wide<32, 32, 32, 4, false>(B, 256, 0, true);
    run_wide<32, 32, 32, 4, true>(B, 512, 512, false);
    run_wide<32, 32, 32, 7, false>(B, 256, 0, true);
    run_wide<32, 32, 32, 7, true>(B, 128, 0, false);
    run_rs<WK_DOWN, 32, 64, 64, 4>(B);
    run_rs<WK_UP, 32, 64, 64, 2>(B);
    run_rs<WK_DOWN, 32, 32, 32, 7>(B);
    run_rs<WK_UP, 32, 32, 32, 4>(B);
    return 0;
}
