// chain.hip — a PERSISTENT LAYER CHAIN of position-tile convolutions (wide.hip) in one launch: consecutive layers of the 512-channel
// L = 2 block of the UNet (down level 5, middle, up level 5: 13 launches of wide_conv_kernel per forward, 22 % of a reverse step)
// run inside ONE kernel, with a barrier among the eight workgroups of a sample tile between layers instead of a kernel boundary.
//
// Why that can be cheaper than the kernel boundary (round 2 measured the opposite for launch chaining ACROSS XCDs: one
// buffer_wbl2 per workgroup cost 11 us): a layer at L = 2 is 8 channel groups x 32 sample tiles = 256 workgroups, and a sample tile's
// next layer needs exactly the eight channel-group outputs of ITS OWN tile.  The chain puts the eight workgroups of a tile (a
// "cluster") on ONE XCD - workgroup ids go round-robin over the 8 XCDs, so ids with equal (id & 7) share an XCD and its L2 - and then
// the L2 is the only coherence point the hand-over needs:
//   producer: plain stores; s_waitcnt vmcnt(0) (the stores have reached the L2); workgroup barrier; ONE L2-local atomic increment of
//             the cluster's counter (workgroup-scope atomic: executed in the XCD's L2, no sc1 = no trip to the memory side);
//   consumer: requests its weight fragments and epilogue operands (they do not depend on anybody), then polls the counter with an
//             L2-local atomic read, workgroup barrier, buffer_inv sc0 (drop this CU's stale L1 lines of the recycled activation
//             buffers), then loads the activations - from the L2 they were just written to, not from HBM.
// No L2 write-back, no L2 invalidate, no dispatch gap, and a cluster never waits for the slowest workgroup of the whole grid.
// Results are bit-identical to the launch-per-layer path: the same wide_conv_body runs on the same tiles.
//
// STATUS (round 4): an EXPERIMENT, opt-in (EDMP_CHAIN=1 when the model is built).  Bit-identical to one launch per layer and x1.00 in
// speed (profiles/r04_l2_chain.md: the cluster gate costs what the kernel boundary costs, ~1-2.5 us; the layers' time is inside the
// kernels).  And it is only valid while NO OTHER kernel of the process starts or ends on the GPU during the chain: with three or more
// other streams busy (row chains of one batch on the same model) stale activation lines were read - the L2-local hand-over does not
// survive the cache maintenance of foreign kernel boundaries.  unet_run_program therefore bypasses chains for row-chained runs.
//
// Safety: all 8 workgroups of a cluster must be resident together.  Clusters are contiguous in dispatch order and a cluster only
// ever waits for its own members, so a chain kernel alone cannot deadlock; against pathological co-tenancy every wait is bounded
// (~0.25 s of shader clock): on expiry the workgroup raises `abort_flag` and leaves, the host discards the result and reruns the layers
// launch by launch.  edmp_ctx probes the id -> XCD mapping once (xcd_probe_kernel) and only enables chains when it is id & 7.
#pragma once
#include "params.h"
#include "wide.hip"

namespace edmp {

// instances a chain op can be (the 512-channel L = 2 block): 0 Conv1dBlock Karatsuba form, 1 the same with the folded residual conv,
// 2 ConvTranspose k4 s2, 3 / 4 the direct-form Conv1dBlock without / with the residual conv (EDMP_NO_KARATSUBA builds)
constexpr int kChainCtrWords = 32;  // counter words per cluster = one 128-byte cache line

enum ChainInst { CI_K5K2 = 0, CI_K5K2_RES = 1, CI_UP = 2, CI_K5 = 3, CI_K5_RES = 4 };

struct ChainP {
    RcbP op[kChainMaxOps];
    int inst[kChainMaxOps];
    int n_ops;
    int n_tiles;          // sample tiles (= clusters) of the batch
    // per cluster ONE 128-byte line: word 0 arrivals of this launch, word 1 workgroups past their last wait (the eighth zeroes both).
    // A line of its own matters: the eight L2s are not coherent with each other and write back whole lines, so counters of clusters
    // on different XCDs in one line would overwrite each other with stale copies whenever another stream's kernel boundary flushes
    // the L2s in the middle of a chain (seen as gates that open early: row chains x layer chains, round 4)
    unsigned* ctr;
    int* abort_flag;      // set by a workgroup whose bounded wait expired (host-visible memory)
};

__device__ __forceinline__ unsigned l2_atomic_read(unsigned* p) {
    unsigned r;
    const unsigned zero = 0;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(zero) : "memory");
    return r;
}
__device__ __forceinline__ void l2_atomic_inc(unsigned* p) {
    const unsigned one = 1;
    asm volatile("global_atomic_add %0, %1, off" ::"v"(p), "v"(one) : "memory");
}

#ifdef EDMP_CHAIN_STAMPS  // tools/chainbench.hip: shader-clock stamps of workgroup 0 per layer: start | gate passed | body done | arrived
__device__ long long g_chain_stamps[kChainMaxOps][4];
#define EDMP_CHAIN_STAMP(k, i) if (blockIdx.x == 0 && threadIdx.x == 0) g_chain_stamps[k][i] = clock64();
#else
#define EDMP_CHAIN_STAMP(k, i)
#endif

// the consumer half of the cluster barrier (wide_conv_body calls it once, see WideNoGate)
struct ChainGate {
    static constexpr bool kGated = true;
    unsigned* ctr;
    unsigned target;  // counter value at which the previous layer of this cluster is complete
    int* abort_flag;
    bool wait;
    int k;
    __device__ __forceinline__ void operator()() const {
        if (wait) {
            if (threadIdx.x == 0) {
                const long long t0 = clock64();
                while ((int)(l2_atomic_read(ctr) - target) < 0) {
                    __builtin_amdgcn_s_sleep(2);
                    if (clock64() - t0 > 600000000ll) {  // ~0.25 s: never in a healthy run
                        *abort_flag = 1;
                        break;
                    }
                }
            }
            __syncthreads();
            asm volatile("buffer_inv sc0" ::: "memory");  // this CU's L1 may hold lines of the recycled buffers from earlier layers
        }
        EDMP_CHAIN_STAMP(k, 1)
    }
};

template <int NOPS_UNUSED = 0>
__global__ __launch_bounds__(256) void l2_chain_kernel(ChainP a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // workgroup id -> (cluster = sample tile, member = channel group): ids with equal (id & 7) sit on one XCD
    const int lin = blockIdx.x;
    const int xcd = lin & 7, j = lin >> 3;
    const int member = j & 7, tile = (j >> 3) * 8 + xcd;
    if (tile >= a.n_tiles) return;
    unsigned* ctr = a.ctr + (size_t)tile * kChainCtrWords;
    for (int k = 0; k < a.n_ops; ++k) {
        const RcbP& p = a.op[k];
        ChainGate gate{ctr, 8u * (unsigned)k, a.abort_flag, k > 0, k};
        EDMP_CHAIN_STAMP(k, 0)
        switch (a.inst[k]) {
            case CI_K5K2: wide_conv_body<WK_K5K2, 32, 64, 64, 2, false>(p, member, tile, lds, gate); break;
            case CI_K5K2_RES: wide_conv_body<WK_K5K2, 32, 64, 64, 2, true>(p, member, tile, lds, gate); break;
            case CI_UP: wide_conv_body<WK_UP, 32, 64, 64, 2, false>(p, member, tile, lds, gate); break;
            case CI_K5: wide_conv_body<WK_K5, 32, 64, 64, 2, false>(p, member, tile, lds, gate); break;
            default: wide_conv_body<WK_K5, 32, 64, 64, 2, true>(p, member, tile, lds, gate); break;
        }
        EDMP_CHAIN_STAMP(k, 2)
        if (k + 1 < a.n_ops) {
            // producer half: this workgroup's stores are in the L2, every wave is past its epilogue (the next layer's prologue
            // writes the LDS the epilogue read), then one arrival per workgroup
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) l2_atomic_inc(ctr);
        }
        EDMP_CHAIN_STAMP(k, 3)
    }
    // leave the counters at zero for the next launch: the cluster's eighth workgroup to get here knows that nobody polls any more
    // (every member has passed its last wait before it counts itself out)
    if (threadIdx.x == 0 && a.n_ops > 1) {
        unsigned* out = ctr + 1;
        unsigned before;
        const unsigned one = 1;
        asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(before) : "v"(out), "v"(one) : "memory");
        if (before == 7u) {
            const unsigned z = 0;
            asm volatile("global_atomic_swap %0, %1, off\n\tglobal_atomic_swap %2, %1, off" ::"v"(ctr), "v"(z), "v"(out) : "memory");
        }
    }
}

constexpr size_t chain_lds_bytes() {
    size_t m = WideCfg<WK_K5K2, 32, 64, 64, 2, false>::lds_bytes();
    const size_t o[] = {WideCfg<WK_K5K2, 32, 64, 64, 2, true>::lds_bytes(), WideCfg<WK_UP, 32, 64, 64, 2, false>::lds_bytes(), WideCfg<WK_K5, 32, 64, 64, 2, false>::lds_bytes(),
                        WideCfg<WK_K5, 32, 64, 64, 2, true>::lds_bytes()};
    for (size_t v : o) m = v > m ? v : m;
    return m;
}

// one launch for `a.n_ops` consecutive layers
int launch_l2_chain(const ChainP& a, hipStream_t s);

// which XCD every workgroup id of a 256-workgroup launch lands on (HW_REG_XCC_ID)
__global__ void xcd_probe_kernel(int* out);
int probe_xcd_round_robin(hipStream_t s, bool* ok);

#ifdef EDMP_CHAIN_DEFINE
int launch_l2_chain(const ChainP& a, hipStream_t s) {
    static std::atomic<int> attr_set{0};
    constexpr size_t bytes = chain_lds_bytes();
    static_assert(bytes <= 160 * 1024, "chain kernel exceeds the 160 KiB LDS of a CU");
    if (!attr_set.load(std::memory_order_acquire)) {
        EDMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&l2_chain_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        attr_set.store(1, std::memory_order_release);
    }
    const int slots = (a.n_tiles + 7) / 8;  // clusters per XCD
    hipLaunchKernelGGL((l2_chain_kernel<0>), dim3(64 * slots), dim3(256), bytes, s, a);
    return EDMP_OK;
}

__global__ void xcd_probe_kernel(int* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf);  // HW_REG_XCC_ID[3:0]
}

// true iff workgroup ids with equal (id & 7) share an XCD and the eight classes sit on eight different XCDs
int probe_xcd_round_robin(hipStream_t s, bool* ok) {
    *ok = false;
    int* d = nullptr;
    EDMP_HIP_CHECK(hipMalloc((void**)&d, 256 * sizeof(int)));
    hipLaunchKernelGGL(xcd_probe_kernel, dim3(256), dim3(64), 0, s, d);
    int h[256];
    hipError_t e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    EDMP_HIP_CHECK(e);
    int cls[8];
    for (int i = 0; i < 8; ++i) cls[i] = h[i];
    bool good = true;
    for (int i = 0; i < 256; ++i) good = good && h[i] == cls[i & 7];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < i; ++j) good = good && cls[i] != cls[j];
    *ok = good;
    return EDMP_OK;
}
#endif

}  // namespace edmp
