// tsp_comm.hip -- multi-GPU image reduce over RCCL / xGMI.
//
// The reference is single-GPU and has no counterpart.  The path shards by particle index range
// (one process per GPU, each with its own tsp_context) and the partial images add, so the only
// exchange step is ONE sum-reduce of the R*R*C float32 render target per frame (8 MiB at
// 1024^2 x 2).  librccl.so is opened lazily with dlopen so single-GPU users never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include "tsp_internal.h"

namespace tsp {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static Rccl g_rccl;

static int load_rccl() {
    if (g_rccl.handle) return TSP_OK;
    void *h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        set_error("cannot load librccl.so: %s", dlerror());
        return TSP_ECOMM;
    }
#define LOAD(field, name)                                               \
    *(void **)(&g_rccl.field) = dlsym(h, name);                         \
    if (!g_rccl.field) {                                                \
        set_error("librccl.so lacks symbol %s", name);                  \
        return TSP_ECOMM;                                               \
    }
    LOAD(GetUniqueId, "ncclGetUniqueId");
    LOAD(CommInitRank, "ncclCommInitRank");
    LOAD(CommDestroy, "ncclCommDestroy");
    LOAD(Reduce, "ncclReduce");
    LOAD(AllReduce, "ncclAllReduce");
    LOAD(GetErrorString, "ncclGetErrorString");
#undef LOAD
    g_rccl.handle = h;
    return TSP_OK;
}

#define TSP_NCCL(call)                                                                   \
    do {                                                                                 \
        ncclResult_t r_ = (call);                                                        \
        if (r_ != ncclSuccess) {                                                         \
            set_error("%s failed: %s", #call, g_rccl.GetErrorString(r_));                \
            return TSP_ECOMM;                                                            \
        }                                                                                \
    } while (0)

}  // namespace tsp

using namespace tsp;

extern "C" {

int tsp_comm_unique_id(char *id_out) {
    TSP_REQUIRE(id_out, TSP_EINVAL, "NULL argument");
    int rc = load_rccl();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == TSP_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    TSP_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return TSP_OK;
}

int tsp_comm_init(tsp_context *ctx, int n_ranks, int rank, const char *id) {
    TSP_REQUIRE(ctx && id, TSP_EINVAL, "NULL argument");
    TSP_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, TSP_EINVAL, "bad rank %d of %d", rank, n_ranks);
    TSP_REQUIRE(!ctx->comm, TSP_ESTATE, "communicator already initialised");
    int rc = load_rccl();
    if (rc) return rc;
    TSP_HIP(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    TSP_NCCL(g_rccl.CommInitRank(&comm, n_ranks, uid, rank));
    ctx->comm = comm;
    ctx->n_ranks = n_ranks;
    ctx->rank = rank;
    return TSP_OK;
}

int tsp_comm_reduce_image(tsp_context *ctx, int root, double *gpu_ms_out) {
    TSP_REQUIRE(ctx, TSP_EINVAL, "NULL context");
    if (gpu_ms_out) *gpu_ms_out = 0.0;
    if (ctx->n_ranks <= 1 && !ctx->comm) return TSP_OK;   // single GPU: the partial image is the image
    TSP_REQUIRE(ctx->comm, TSP_ESTATE, "tsp_comm_init has not been called");
    TSP_REQUIRE(root < ctx->n_ranks, TSP_EINVAL, "root %d out of range", root);
    // The reduce runs in place on the float32 presentation copy; the float64 accumulator stays local.  Reducing the
    // same frame twice would add the other ranks' shares twice on the root, so it is refused: render (or
    // tsp_write_image) first, which rebuilds the local partial image.
    TSP_REQUIRE(!ctx->image_is_reduced, TSP_ESTATE,
                "the render target was already reduced for this frame (call tsp_render before reducing again)");
    TSP_HIP(hipSetDevice(ctx->device));
    const size_t count = (size_t)ctx->R * ctx->R * ctx->C;
    TSP_HIP(hipEventRecord(ctx->ev[4], ctx->stream));
    if (root < 0)
        TSP_NCCL(g_rccl.AllReduce(ctx->image, ctx->image, count, ncclFloat, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
    else
        TSP_NCCL(g_rccl.Reduce(ctx->image, ctx->image, count, ncclFloat, ncclSum, root, (ncclComm_t)ctx->comm, ctx->stream));
    TSP_HIP(hipEventRecord(ctx->ev[5], ctx->stream));
    TSP_HIP(hipStreamSynchronize(ctx->stream));
    ctx->image_is_reduced = true;
    float ms = 0.f;
    TSP_HIP(hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]));
    if (gpu_ms_out) *gpu_ms_out = ms;
    return TSP_OK;
}

int tsp_comm_destroy(tsp_context *ctx) {
    if (!ctx || !ctx->comm) return TSP_OK;
    if (g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->n_ranks = 1;
    ctx->rank = 0;
    return TSP_OK;
}

}  // extern "C"
