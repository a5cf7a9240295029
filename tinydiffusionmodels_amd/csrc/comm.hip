// Data-parallel collective of the train step behind the C ABI: one
// process per GPU, ONE all-reduce(SUM) of the flat fp32 gradient per step over RCCL / xGMI,
// enqueued on the caller's stream (so it orders after the backward kernels and before AdamW
// without a host sync, and is capturable into a hipGraph).  The reference has no distributed
// code (deployment/configs/mnist-training.yaml:5-6 trains on one device); this layer is new.
//
// RCCL is bound at run time (dlopen of librccl.so.1): the single-GPU paths — and the plain-C
// host of tests/c_abi — never load it, and inside a PyTorch process the already-loaded copy
// of the same SONAME is reused, so there is exactly one RCCL in the process.
#include "tdm_common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <mutex>
#include <string.h>

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle != nullptr) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h != nullptr) break;
    }
    if (h == nullptr) {
        tdm_set_error("comm: cannot load RCCL (librccl.so.1): %s", dlerror());
        return 1;
    }
    Rccl r;
    r.handle = h;
#define TDM_SYM(field, name)                                                       \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                 \
    if (r.field == nullptr) {                                                      \
        tdm_set_error("comm: RCCL symbol %s missing: %s", name, dlerror());        \
        return 1;                                                                  \
    }
    TDM_SYM(GetUniqueId, "ncclGetUniqueId")
    TDM_SYM(CommInitRank, "ncclCommInitRank")
    TDM_SYM(CommDestroy, "ncclCommDestroy")
    TDM_SYM(AllReduce, "ncclAllReduce")
    TDM_SYM(Broadcast, "ncclBroadcast")
    TDM_SYM(GetErrorString, "ncclGetErrorString")
    TDM_SYM(GetVersion, "ncclGetVersion")
#undef TDM_SYM
    g_rccl = r;
    return 0;
}

#define TDM_NCCL(call, what)                                                                   \
    do {                                                                                       \
        ncclResult_t r__ = (call);                                                             \
        if (r__ != ncclSuccess) {                                                              \
            tdm_set_error("comm: %s failed: %s", what, g_rccl.GetErrorString(r__));            \
            return 200 + (int)r__;                                                             \
        }                                                                                      \
    } while (0)

}  // namespace

extern "C" {

int tdm_ctx_create(int device, tdm_ctx** out) {
    TDM_REQUIRE(out != nullptr && device >= 0, "ctx_create: bad arguments");
    tdm_ctx* c = new tdm_ctx();   // default arithmetic, overlap on, no communicator, no side queue yet (tdm_common.h)
    c->device = device;
    *out = c;
    return 0;
}

int tdm_ctx_destroy(tdm_ctx* ctx) {
    if (ctx == nullptr) return 0;
    TDM_TRY(tdm_ctx_unbind_for_destroy(ctx));
    int rc = 0;
    if (ctx->comm != nullptr && g_rccl.handle != nullptr) {
        ncclResult_t r = g_rccl.CommDestroy((ncclComm_t)ctx->comm);
        if (r != ncclSuccess) {
            tdm_set_error("comm: ncclCommDestroy failed: %s", g_rccl.GetErrorString(r));
            rc = 200 + (int)r;
        }
    }
    delete ctx;
    return rc;
}

int tdm_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

int tdm_comm_unique_id(void* out_bytes) {
    TDM_REQUIRE(out_bytes != nullptr, "comm_unique_id: NULL buffer");
    TDM_TRY(load_rccl());
    ncclUniqueId id;
    TDM_NCCL(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(out_bytes, &id, sizeof(id));
    return 0;
}

int tdm_comm_init(tdm_ctx* ctx, const void* unique_id, int rank, int world) {
    TDM_REQUIRE(ctx != nullptr && unique_id != nullptr, "comm_init: NULL argument");
    TDM_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init: rank %d of world %d", rank, world);
    TDM_REQUIRE(ctx->comm == nullptr, "comm_init: communicator already initialised");
    TDM_TRY(load_rccl());
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) {
        tdm_set_error("comm_init: hipSetDevice(%d): %s", ctx->device, hipGetErrorString(e));
        return 100 + (int)e;
    }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    TDM_NCCL(g_rccl.CommInitRank(&comm, world, id, rank), "ncclCommInitRank");
    ctx->comm = comm; ctx->rank = rank; ctx->world = world;
    return 0;
}

int tdm_comm_rank(const tdm_ctx* ctx) { return ctx ? ctx->rank : 0; }
int tdm_comm_world(const tdm_ctx* ctx) { return ctx ? ctx->world : 1; }

int tdm_comm_rccl_version(void) {
    if (load_rccl() != 0) return -1;
    int v = 0;
    if (g_rccl.GetVersion(&v) != ncclSuccess) return -1;
    return v;
}

int tdm_allreduce_sum_f32(tdm_ctx* ctx, float* buf, int64_t n, void* stream) {
    TDM_REQUIRE(ctx != nullptr && ctx->comm != nullptr, "allreduce: communicator not initialised (tdm_comm_init)");
    TDM_REQUIRE(buf != nullptr && n > 0, "allreduce: empty buffer");
    TDM_NCCL(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, (ncclComm_t)ctx->comm, (hipStream_t)stream), "ncclAllReduce");
    return 0;
}

int tdm_broadcast_f32(tdm_ctx* ctx, float* buf, int64_t n, int root, void* stream) {
    TDM_REQUIRE(ctx != nullptr && ctx->comm != nullptr, "broadcast: communicator not initialised (tdm_comm_init)");
    TDM_REQUIRE(buf != nullptr && n > 0 && root >= 0 && root < ctx->world, "broadcast: bad arguments");
    TDM_NCCL(g_rccl.Broadcast(buf, buf, (size_t)n, ncclFloat32, root, (ncclComm_t)ctx->comm, (hipStream_t)stream), "ncclBroadcast");
    return 0;
}

}  // extern "C"
