// common.h — context, error handling and small device helpers shared by the libedmp_hip translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include <map>
#include <unordered_map>
#include <vector>

#include "../../include/edmp_hip.h"

namespace edmp {

void set_error(const char* fmt, ...);

#define EDMP_HIP_CHECK(expr)                                                                      \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            edmp::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return EDMP_ERR_HIP;                                                                  \
        }                                                                                         \
    } while (0)

#define EDMP_REQUIRE(cond, ...)           \
    do {                                  \
        if (!(cond)) {                    \
            edmp::set_error(__VA_ARGS__); \
            return EDMP_ERR_ARG;          \
        }                                 \
    } while (0)

struct UNet;     // unet.hip
struct Guide;    // guide.hip
struct Sampler;  // sampler.hip

// per-launch timing of the UNet layer program (HIP events on the context's stream around every conv-family launch),
// switched on by edmp_prof_enable for bench.py's roofline pass; folded per program op by prof_fold()
struct Prof {
    int on = 0;  // 0 off | 1 one event pair per conv launch (per-op table) | 2 one pair around the whole layer program (total)
    struct Pend {
        hipEvent_t a, b;
        int op;
    };
    std::vector<Pend> pending;  // recorded, not yet read
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    std::vector<double> op_ms;      // per program op
    std::vector<int64_t> op_calls;
    double conv_ms = 0.0;
    int64_t conv_launches = 0;
};

// Hot-path Mish: with n = e^x, tanh(log(1+n)) = (n^2+2n)/(n^2+2n+2) exactly, so one v_exp_f32 and one v_rcp_f32
// replace the expf/log1pf/tanhf library chain (~15x fewer VALU instructions; it dominated the GroupNorm kernels).
// Relative error ~3e-7 (1-ulp exp2/rcp), same class as torch's own fp32 Mish.  Above torch's softplus threshold (x > 20,
// where torch returns x) the clamp makes w = n (n + 2) ~ 2e17, w + 2 == w and the ratio 1 to within the same ulp: no
// select needed (a compare + select per element is 7 % of the epilogue's VALU work).
__device__ __forceinline__ float mish_fast(float x) {
    const float n = __builtin_amdgcn_exp2f(fminf(x, 20.0f) * 1.44269504088896340736f);
    const float w = n * (n + 2.0f);
    return x * (w * __builtin_amdgcn_rcpf(w + 2.0f));
}

}  // namespace edmp

struct edmp_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    edmp::UNet* unet = nullptr;    // the CURRENT model (slot unet_key)
    edmp::Guide* guide = nullptr;  // the CURRENT scene + rows (slot guide_key)
    // resident, not current, most recently used first: a caller that alternates between a few models / per-scene guides
    // (the reference keeps one guide object per scene, infer_serial.py:112) switches by key instead of re-uploading
    // 120 MB of weights or rebuilding the obstacle table
    uint64_t unet_key = 0, guide_key = 0;
    std::vector<std::pair<uint64_t, edmp::UNet*>> unet_slots;
    std::vector<std::pair<uint64_t, edmp::Guide*>> guide_slots;
    int unet_cap = 3, guide_cap = 8;  // resident objects per context, the current one included
    edmp::Sampler* sampler = nullptr;
    edmp::Prof prof;
    uint64_t epoch = 0;  // bumped whenever device pointers / tables a captured hipGraph baked in may have changed
    // recycled device blocks of the per-scene objects (guide tables, row arrays, scratch): hipFree waits for EVERY stream of the
    // device, so with two scenes in flight each of the ~20 frees of a scene change stalled its host thread behind the other scene's
    // queued loop (72 ms per scene, profiles/r05_problem_set.md).  Blocks go back here instead and are handed out again by capacity;
    // everything is freed when the context is destroyed.  Callers only recycle blocks no enqueued work still reads (they
    // synchronise the context's stream first: edmp_guide_slot).
    std::multimap<size_t, void*> pool_free;
    std::unordered_map<void*, size_t> pool_cap;
    size_t pool_bytes = 0;
    int* d_int = nullptr;  // one device int for small read-backs (edmp_argmin_dev)
};

namespace edmp {

void unet_destroy(UNet*);
bool unet_complete(const UNet*);   // loaded: has a layer program
bool guide_complete(const Guide*);  // scene tables and row arrays both set
int prof_fold(edmp_ctx* ctx);  // unet.hip: read the pending event pairs into the per-op / total accumulators
void guide_destroy(edmp_ctx* ctx, Guide*);  // the device blocks go back to the context's pool
int ctx_alloc(edmp_ctx* ctx, void** p, size_t bytes);  // guide.hip: from the context's block pool, else hipMalloc
void ctx_release(edmp_ctx* ctx, void* p);               // back into the pool (hipFree only beyond the pool's byte cap)
void ctx_pool_destroy(edmp_ctx* ctx);
void sampler_destroy(Sampler*);

// RAII-less helper: device allocation tracked by the owner
template <class T>
inline hipError_t dev_alloc(T** p, size_t n) {
    return hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T));
}

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// x + (x of lane ^ M) for M in {1, 2, 4, 8} on the DPP data path (a few cycles) instead of __shfl_xor's ds_bpermute
// (an LDS round trip, ~100 cycles each; six of them sat on the critical path of every GroupNorm epilogue).
// M = 1, 2: quad permutes; M = 8: rotate the 16-lane row by 8; M = 4: mirror the 8-lane half-row, which equals
// lane ^ 4 only for values already uniform within quads - i.e. apply the steps in the order 1, 2, 4, (8).
template <int M>
__device__ __forceinline__ float dpp_xor_add(float x) {
    static_assert(M == 1 || M == 2 || M == 4 || M == 8, "lane-xor distance within a 16-lane row");
    constexpr int ctrl = (M == 1) ? 0xB1 : (M == 2) ? 0x4E : (M == 4) ? 0x141 : 0x128;
    const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xF, 0xF, true);
    return x + __builtin_bit_cast(float, y);
}

// x + (x of lane ^ 16) resp. (lane ^ 32) with gfx950's v_permlane16_swap / v_permlane32_swap (VALU, a few cycles) instead of
// a ds_bpermute round trip: swapping (x, x) leaves the even half-rows' value in one result register and the odd
// half-rows' in the other, for every lane
__device__ __forceinline__ float swap16_add(float x) {
    const int xi = __builtin_bit_cast(int, x);
    const auto r = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);
    return __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
}
__device__ __forceinline__ float swap32_add(float x) {
    const int xi = __builtin_bit_cast(int, x);
    const auto r = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
    return __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
}

// two elements at once: the non-transcendental half of the formula on packed-fp32 instructions (v_pk_mul/add/fma_f32: two
// lanes' worth of work per issue slot); per component the same operations in the same order as mish_fast
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t mish_fast2(f32x2_t x) {
    f32x2_t xm = {fminf(x.x, 20.0f), fminf(x.y, 20.0f)};
    xm = xm * 1.44269504088896340736f;
    const f32x2_t n = {__builtin_amdgcn_exp2f(xm.x), __builtin_amdgcn_exp2f(xm.y)};
    const f32x2_t w = n * (n + 2.0f);
    const f32x2_t d = w + 2.0f;
    const f32x2_t rc = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    return x * (w * rc);
}

// a + b on both halves in ONE instruction.  Plain `a + b` on two-element vectors that were extracted from a float4 and go back
// into one is split into scalar adds by the compiler; inside the conv K loop every VALU instruction costs an MFMA issue slot
__device__ __forceinline__ f32x2_t pk_add(f32x2_t a, f32x2_t b) {
    f32x2_t r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// x * tanh(softplus(x)) with torch's softplus threshold (20): blocks.py:27,65 -> torch.nn.Mish
__device__ __forceinline__ float mish_f(float x) {
    float sp = (x > 20.0f) ? x : log1pf(expf(x));
    return x * tanhf(sp);
}

}  // namespace edmp
