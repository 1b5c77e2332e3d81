// Row-shard collective of the GMM-VB data pass behind the C ABI: one in-place all-reduce(sum, f64) of the statistics
// block [ns | h | a | B] per VB iteration over RCCL (xGMI between the GPUs of a node).  The reference has no
// distributed code; a binding that shards rows across processes needs nothing but these four calls
// and a way to hand 128 bytes from rank 0 to the other ranks.
//
// RCCL is resolved at run time (dlopen "librccl.so.1": the copy the process already uses, e.g. PyTorch's, or ROCm's),
// so the library still loads - and every other entry point works - on a machine without it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <new>

#include "workspace.h"

using namespace gmmvb;

struct gmmvb_comm {
    ncclComm_t comm = nullptr;
    int n_ranks = 0, rank = 0;
};

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::once_flag g_once;

const Rccl* rccl() {
    std::call_once(g_once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.handle) break;
        }
        if (!g_rccl.handle) return;
        g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(g_rccl.handle, "ncclGetUniqueId");
        g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(g_rccl.handle, "ncclCommInitRank");
        g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(g_rccl.handle, "ncclCommDestroy");
        g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(g_rccl.handle, "ncclAllReduce");
        g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(g_rccl.handle, "ncclGetErrorString");
    });
    const bool ok = g_rccl.handle && g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce;
    return ok ? &g_rccl : nullptr;
}

int nccl_fail(const Rccl* r, const char* what, ncclResult_t rc) {
    static thread_local char msg[256];
    std::snprintf(msg, sizeof(msg), "%s: %s", what, r->GetErrorString ? r->GetErrorString(rc) : "RCCL error");
    return fail(GMMVB_EHIP, msg);
}

}  // namespace

extern "C" {

int gmmvb_comm_unique_id(unsigned char* id_out /*[128]*/) {
    if (!id_out) return fail(GMMVB_EINVAL, "null argument");
    const Rccl* r = rccl();
    if (!r) return fail(GMMVB_EUNSUPPORTED, "librccl.so.1 could not be loaded");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    ncclResult_t rc = r->GetUniqueId(&id);
    if (rc != ncclSuccess) return nccl_fail(r, "ncclGetUniqueId", rc);
    std::memcpy(id_out, &id, sizeof(id));
    return GMMVB_OK;
}

int gmmvb_comm_create(const unsigned char* id /*[128]*/, int n_ranks, int rank, gmmvb_comm** out) {
    if (!id || !out) return fail(GMMVB_EINVAL, "null argument");
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(GMMVB_EINVAL, "rank must be in [0, n_ranks)");
    const Rccl* r = rccl();
    if (!r) return fail(GMMVB_EUNSUPPORTED, "librccl.so.1 could not be loaded");
    gmmvb_comm* c = new (std::nothrow) gmmvb_comm();
    if (!c) return fail(GMMVB_ENOMEM, "host allocation failed");
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclResult_t rc = r->CommInitRank(&c->comm, n_ranks, uid, rank);      // collective: every rank calls it
    if (rc != ncclSuccess) {
        delete c;
        return nccl_fail(r, "ncclCommInitRank", rc);
    }
    c->n_ranks = n_ranks;
    c->rank = rank;
    *out = c;
    return GMMVB_OK;
}

int gmmvb_comm_destroy(gmmvb_comm* comm) {
    if (!comm) return GMMVB_OK;
    const Rccl* r = rccl();
    if (r && comm->comm) (void)r->CommDestroy(comm->comm);
    delete comm;
    return GMMVB_OK;
}

int gmmvb_allreduce_stats(gmmvb_comm* comm, double* stats_dev, int64_t len, void* stream) {
    if (!comm || !stats_dev || len < 1) return fail(GMMVB_EINVAL, "bad argument");
    const Rccl* r = rccl();
    if (!r) return fail(GMMVB_EUNSUPPORTED, "librccl.so.1 could not be loaded");
    ncclResult_t rc = r->AllReduce(stats_dev, stats_dev, (size_t)len, ncclFloat64, ncclSum, comm->comm, (hipStream_t)stream);
    if (rc != ncclSuccess) return nccl_fail(r, "ncclAllReduce", rc);
    return GMMVB_OK;
}

}  // extern "C"
