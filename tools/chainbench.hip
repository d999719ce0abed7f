// chainbench.hip — the persistent layer chain (edmp_amd/csrc/chain.hip) against one launch per layer on the same layers: n Conv1dBlocks
// 512 -> 512 at L = 2 (Karatsuba form) ping-ponging between two activation buffers, B = 1024.  Checks bit-equality of the final
// activations, times both (chains of 20 repetitions, best of 6) and prints the per-layer shader-clock stamps of workgroup 0 of the chain
// kernel: wait in the cluster gate | body | arrive.
// (round 5: chain.hip left the library; `git apply tools/experiments/r04_l2_chain.patch` restores it for this tool)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=12 tools/chainbench.hip -o tools/chainbench && tools/chainbench [n = 13]
#define EDMP_CHAIN_STAMPS 1
#define EDMP_CHAIN_DEFINE 1
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
#include "../edmp_amd/csrc/chain.hip"
using namespace edmp;

template <class T>
static T* up(const std::vector<T>& h) {
    T* d;
    hipMalloc((void**)&d, h.size() * sizeof(T));
    hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 13;
    const int B = 1024, C = 512, L = 2;
    std::mt19937 g(1);
    std::uniform_real_distribution<float> d(-1.f, 1.f);
    std::normal_distribution<float> nd(0.f, 1.f);
    const float ws = 1.7f / std::sqrt(5.0f * C);
    std::vector<float> hx((size_t)B * L * C), hb(C), hg(C), hbe(C), htb(C);
    for (auto& v : hx) v = nd(g);
    for (auto& v : hb) v = 0.1f * d(g);
    for (auto& v : hg) v = 1.f + 0.5f * d(g);
    for (auto& v : hbe) v = 0.3f * d(g);
    for (auto& v : htb) v = 0.5f * d(g);
    using C32 = WideCfg<WK_K5K2, 32, 64, 64, 2, false>;
    std::vector<float*> W(n);
    for (int k = 0; k < n; ++k) {
        std::vector<float> hW((size_t)6 * C * C, 0.f), hWf((size_t)(C / 32) * (C / 8) * C32::NSLAB * 256);
        for (size_t i = 0; i < (size_t)5 * C * C; ++i) hW[i] = ws * d(g);
        pack_fragments_k2(hW.data(), C, C, false, hWf.data());
        W[k] = up(hWf);
    }
    float *x0 = up(hx), *bias = up(hb), *gam = up(hg), *bet = up(hbe), *tb = up(htb), *buf[2][2];
    const size_t nel = (size_t)B * L * C;
    for (int v = 0; v < 2; ++v)
        for (int i = 0; i < 2; ++i) hipMalloc((void**)&buf[v][i], nel * 4);
    auto op = [&](int v, int k) {
        RcbP p{};
        p.src1 = k == 0 ? x0 : buf[v][(k - 1) & 1];
        p.C1 = C, p.W = W[k], p.bias = bias, p.gamma = gam, p.beta = bet, p.add_tb = tb, p.dst = buf[v][k & 1], p.Cout = C, p.B = B;
        return p;
    };
    unsigned* ctr;
    hipMalloc((void**)&ctr, 32 * kChainCtrWords * sizeof(unsigned));
    hipMemset(ctr, 0, 32 * kChainCtrWords * sizeof(unsigned));
    int* flag;
    hipHostMalloc((void**)&flag, sizeof(int), hipHostMallocMapped);
    *flag = 0;
    ChainP a{};
    for (int k = 0; k < n; ++k) a.op[k] = op(1, k), a.inst[k] = CI_K5K2;
    a.n_ops = n, a.n_tiles = B / 32, a.ctr = ctr, a.abort_flag = flag;
    auto per_layer = [&]() { for (int k = 0; k < n; ++k) launch_wide_t<WK_K5K2, 32, 64, 64, 2, false>(op(0, k), 0); };
    per_layer();
    launch_l2_chain(a, 0);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess || *flag) { printf("failed: %s, abort flag %d\n", hipGetErrorString(e), *flag); return 1; }
    std::vector<float> y0(nel), y1(nel);
    hipMemcpy(y0.data(), buf[0][(n - 1) & 1], nel * 4, hipMemcpyDeviceToHost);
    hipMemcpy(y1.data(), buf[1][(n - 1) & 1], nel * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    double rms = 0;
    for (size_t i = 0; i < nel; ++i) bad += y0[i] != y1[i], rms += (double)y0[i] * y0[i];
    printf("%d layers: chain output == per-layer output in %zu of %zu elements (rms %.3f)\n", n, nel - bad, nel, std::sqrt(rms / nel));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float bl = 1e9f, bc = 1e9f;
    for (int r = 0; r < 6; ++r) {
        float ms;
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) per_layer();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        bl = std::min(bl, ms * 50.f);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) launch_l2_chain(a, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        bc = std::min(bc, ms * 50.f);
    }
    printf("us per %d layers: one launch per layer %.1f (%.2f per layer) | one chain launch %.1f (%.2f per layer) | x%.3f\n", n, bl, bl / n, bc, bc / n, bl / bc);
    long long st[kChainMaxOps][4];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(edmp::g_chain_stamps), sizeof(st));
    for (int k = 0; k < n; ++k)
        printf("  layer %2d (workgroup 0): until the gate opened %6lld | rest of the body %6lld | drain + arrive %5lld | cycles\n", k, st[k][1] - st[k][0], st[k][2] - st[k][1], st[k][3] - st[k][2]);
    return *flag;
}
