// rccl_hook.hip — the per-guided-step all-reduce of the "one logical batch over several GPUs" mode as NATIVE code: one
// ncclAllReduce of the f64 device scalar sum(g^2) on the context's stream, no Python / GIL on the path of the device-resident loop.
//
// The reference has no distributed code; its only coupling between the rows of a batch is gradient1 / np.linalg.norm(gradient1)
// (/root/reference/lib/guide.py:629).  edmp_sampler_set_allreduce (sampler.hip) takes any C function for that sum; until round 6
// the only one was a Python ctypes callback into torch.distributed (24 us of host time x 125 guided steps at N = 1).  This file
// provides the hook itself.  RCCL is NOT a link dependency of libedmp_hip.so: the six entry points are looked up at run time in
// the RCCL the process already holds (the one libtorch_hip.so brought in, when the host is Python) or in a library the host names,
// so a second copy of RCCL never enters the process.  Types are declared here from rccl.h's published ABI (ncclUniqueId = 128
// opaque bytes by value, ncclFloat64 = 8, ncclSum = 0, ncclSuccess = 0).
//
// Two ways to obtain the communicator: edmp_rccl_attach (own communicator from a unique id the ranks exchanged by any means - one
// rank per GPU, ncclCommInitRank on the context's device) and edmp_rccl_attach_comm (borrow one the host already has, e.g.
// torch's ProcessGroupNCCL._comm_ptr()).  ncclAllReduce is stream-capturable, so whole-run hipGraph replay stays legal with this
// hook installed (it is not with an arbitrary caller hook).
#include <dlfcn.h>

#include <mutex>

namespace edmp {

struct NcclId {
    char internal[128];
};
struct RcclApi {
    void* so = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok() const { return AllReduce != nullptr; }
};
static RcclApi g_rccl;
static std::mutex g_rccl_mutex;  // edmp_rccl_load may be reached from several host threads (one context per scene in flight)

struct RcclHook {
    void* comm = nullptr;
    bool owned = false;
    int nranks = 0, rank = 0;
};

static const char* rccl_err(int rc) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"; }

#define EDMP_RCCL_CHECK(expr)                                                                  \
    do {                                                                                       \
        int _r = (expr);                                                                       \
        if (_r != 0) {                                                                         \
            edmp::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, edmp::rccl_err(_r)); \
            return EDMP_ERR_STATE;                                                             \
        }                                                                                      \
    } while (0)

// the hook proper: what the device-resident loop calls once per guided step (edmp_allreduce_fn)
static int rccl_allreduce_hook(void* user, void* stream, double* sumsq_dev) {
    RcclHook* h = (RcclHook*)user;
    return g_rccl.AllReduce(sumsq_dev, sumsq_dev, 1, /*ncclFloat64*/ 8, /*ncclSum*/ 0, h->comm, (hipStream_t)stream);
}

static void rccl_hook_drop(Sampler* s) {
    RcclHook* h = s->rccl;
    if (s->ar_fn == rccl_allreduce_hook) s->ar_fn = nullptr, s->ar_user = nullptr;
    if (!h) return;
    if (h->owned && h->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(h->comm);
    delete h;
    s->rccl = nullptr;
}

void sampler_rccl_destroy(Sampler* s) { rccl_hook_drop(s); }
bool sampler_hook_is_native(const Sampler* s) { return s->ar_fn == rccl_allreduce_hook; }

static int rccl_install(edmp_ctx* ctx, RcclHook* h) {
    Sampler* s = ctx->sampler;
    rccl_hook_drop(s);
    s->rccl = h;
    s->ar_fn = rccl_allreduce_hook;
    s->ar_user = h;
    return EDMP_OK;
}

}  // namespace edmp

extern "C" int edmp_rccl_load(const char* path) {
    using namespace edmp;
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.ok() && !(path && path[0])) return EDMP_OK;
    void* so = nullptr;
    if (path && path[0]) {
        so = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
        EDMP_REQUIRE(so, "edmp_rccl_load: dlopen(%s) failed: %s", path, dlerror());
    } else {
        // the RCCL already in the process first (a host that runs torch.distributed has exactly one), then the system's
        for (const char* name : {"librccl.so", "librccl.so.1"})
            if (!so) so = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        if (!so && dlsym(RTLD_DEFAULT, "ncclAllReduce")) so = dlopen(nullptr, RTLD_NOW);
        if (!so) so = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        EDMP_REQUIRE(so, "edmp_rccl_load: no RCCL in the process and librccl.so.1 not loadable: %s", dlerror());
    }
    RcclApi a;
    a.so = so;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(so, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(so, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(so, "ncclCommDestroy");
    a.CommCount = (decltype(a.CommCount))dlsym(so, "ncclCommCount");
    a.CommUserRank = (decltype(a.CommUserRank))dlsym(so, "ncclCommUserRank");
    a.AllReduce = (decltype(a.AllReduce))dlsym(so, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(so, "ncclGetErrorString");
    EDMP_REQUIRE(a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.CommCount && a.CommUserRank && a.AllReduce,
                 "edmp_rccl_load: %s does not export the nccl* entry points", path && path[0] ? path : "the RCCL found");
    g_rccl = a;
    return EDMP_OK;
}

extern "C" int edmp_rccl_unique_id(void* id128) {
    using namespace edmp;
    EDMP_REQUIRE(id128, "edmp_rccl_unique_id: null pointer");
    if (!g_rccl.ok()) {
        int rc = edmp_rccl_load(nullptr);
        if (rc) return rc;
    }
    NcclId id;
    EDMP_RCCL_CHECK(g_rccl.GetUniqueId(&id));
    memcpy(id128, id.internal, sizeof id.internal);
    return EDMP_OK;
}

extern "C" int edmp_rccl_attach(edmp_ctx* ctx, const void* id128, int nranks, int rank) {
    using namespace edmp;
    EDMP_REQUIRE(ctx && ctx->sampler, "edmp_rccl_attach: sampler not initialised");
    EDMP_REQUIRE(id128 && nranks >= 1 && rank >= 0 && rank < nranks, "edmp_rccl_attach: bad arguments (nranks %d, rank %d)", nranks, rank);
    if (!g_rccl.ok()) {
        int rc = edmp_rccl_load(nullptr);
        if (rc) return rc;
    }
    EDMP_HIP_CHECK(hipSetDevice(ctx->device));
    EDMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    NcclId id;
    memcpy(id.internal, id128, sizeof id.internal);
    void* comm = nullptr;
    EDMP_RCCL_CHECK(g_rccl.CommInitRank(&comm, nranks, id, rank));
    RcclHook* h = new RcclHook;
    h->comm = comm, h->owned = true, h->nranks = nranks, h->rank = rank;
    return rccl_install(ctx, h);
}

extern "C" int edmp_rccl_attach_comm(edmp_ctx* ctx, void* nccl_comm) {
    using namespace edmp;
    EDMP_REQUIRE(ctx && ctx->sampler, "edmp_rccl_attach_comm: sampler not initialised");
    EDMP_REQUIRE(nccl_comm, "edmp_rccl_attach_comm: null communicator");
    if (!g_rccl.ok()) {
        int rc = edmp_rccl_load(nullptr);
        if (rc) return rc;
    }
    RcclHook* h = new RcclHook;
    h->comm = nccl_comm, h->owned = false;
    int rc1 = g_rccl.CommCount(nccl_comm, &h->nranks), rc2 = g_rccl.CommUserRank(nccl_comm, &h->rank);
    if (rc1 || rc2) {
        delete h;
        set_error("edmp_rccl_attach_comm: %p is not a communicator of the RCCL in this process (%s)", nccl_comm, rccl_err(rc1 ? rc1 : rc2));
        return EDMP_ERR_ARG;
    }
    return rccl_install(ctx, h);
}

extern "C" int edmp_rccl_detach(edmp_ctx* ctx) {
    using namespace edmp;
    EDMP_REQUIRE(ctx && ctx->sampler, "edmp_rccl_detach: sampler not initialised");
    if (ctx->sampler->rccl) {
        EDMP_HIP_CHECK(hipSetDevice(ctx->device));
        EDMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // the last all-reduce must have left the communicator
        rccl_hook_drop(ctx->sampler);
    }
    return EDMP_OK;
}

extern "C" int edmp_rccl_enable(edmp_ctx* ctx, int on) {
    using namespace edmp;
    EDMP_REQUIRE(ctx && ctx->sampler, "edmp_rccl_enable: sampler not initialised");
    Sampler* s = ctx->sampler;
    EDMP_REQUIRE(s->rccl, "edmp_rccl_enable: no communicator attached (edmp_rccl_attach / edmp_rccl_attach_comm first)");
    EDMP_REQUIRE(!s->ar_fn || s->ar_fn == rccl_allreduce_hook, "edmp_rccl_enable: a caller hook is installed (edmp_sampler_set_allreduce)");
    s->ar_fn = on ? rccl_allreduce_hook : nullptr;
    s->ar_user = on ? s->rccl : nullptr;
    return EDMP_OK;
}

extern "C" int edmp_rccl_info(edmp_ctx* ctx, int32_t out[3]) {
    using namespace edmp;
    EDMP_REQUIRE(ctx && ctx->sampler && out, "edmp_rccl_info: bad arguments");
    const RcclHook* h = ctx->sampler->rccl;
    out[0] = h ? h->nranks : 0, out[1] = h ? h->rank : 0, out[2] = h ? (h->owned ? 1 : 2) : 0;  // (attached; edmp_rccl_enable switches it on / off)
    return EDMP_OK;
}

extern "C" int edmp_sampler_allreduce_stats(edmp_ctx* ctx, uint64_t out[3], int reset) {
    EDMP_REQUIRE(ctx && ctx->sampler && out, "edmp_sampler_allreduce_stats: bad arguments");
    edmp::Sampler* s = ctx->sampler;
    out[0] = s->ar_calls, out[1] = s->ar_ns, out[2] = s->ar_max_ns;
    if (reset) s->ar_calls = s->ar_ns = s->ar_max_ns = 0;
    return EDMP_OK;
}
