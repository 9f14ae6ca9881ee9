// Gradient all-reduce over RCCL / xGMI on a side HIP stream (reference: train.py:113,182 DistributedDataParallel over NCCL).
// One communicator per process (one process per GPU).  RCCL is NOT linked: the host names the librccl to use
// (maed_comm_load; the Python side passes the copy PyTorch-ROCm already has in the process) and the five entry points
// are bound with dlsym, so libmaed_hip.so loads on a box without RCCL and never mixes two RCCL builds.
//
// Ordering protocol (all asynchronous, no host sync):
//   maed_comm_allreduce_async(buf, .., compute_stream): event on compute_stream -> side stream waits -> ncclAllReduce(SUM)
//       on the side stream.  Called once per gradient bucket as soon as backward has produced it, so the collective
//       overlaps the rest of backward.
//   maed_comm_wait(compute_stream): event on the side stream -> compute_stream waits.  Called once before the optimizer.
#include "common.cuh"
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

namespace {
struct Api {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
} g_api;
ncclComm_t g_comm = nullptr;
hipStream_t g_side = nullptr;
hipEvent_t g_ev_ready = nullptr, g_ev_done = nullptr;
int g_world = 0;

#define COMM_HIP(expr, what) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { \
    maed_set_error("%s: %s", what, hipGetErrorString(e__)); return MAED_ERR_LAUNCH; } } while (0)
#define COMM_NCCL(expr, what) do { ncclResult_t r__ = (expr); if (r__ != ncclSuccess) { \
    maed_set_error("%s: %s", what, g_api.GetErrorString ? g_api.GetErrorString(r__) : "rccl error"); return MAED_ERR_LAUNCH; } } while (0)
}  // namespace

extern "C" int maed_comm_load(const char* librccl_path) {
    if (g_api.handle) return MAED_OK;
    void* h = dlopen(librccl_path ? librccl_path : "librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    MAED_CHECK_ARG(h, MAED_ERR_UNSUPPORTED, "comm_load: dlopen(%s) failed: %s", librccl_path ? librccl_path : "librccl.so.1", dlerror());
    g_api.GetUniqueId = (decltype(g_api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_api.CommInitRank = (decltype(g_api.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_api.AllReduce = (decltype(g_api.AllReduce))dlsym(h, "ncclAllReduce");
    g_api.CommDestroy = (decltype(g_api.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_api.GetErrorString = (decltype(g_api.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!(g_api.GetUniqueId && g_api.CommInitRank && g_api.AllReduce && g_api.CommDestroy)) {
        maed_set_error("comm_load: %s does not export the NCCL API", librccl_path ? librccl_path : "librccl.so.1");
        dlclose(h);
        g_api = Api();
        return MAED_ERR_UNSUPPORTED;
    }
    g_api.handle = h;
    return MAED_OK;
}

extern "C" int maed_comm_unique_id(void* id128) {
    MAED_CHECK_ARG(g_api.handle, MAED_ERR_UNSUPPORTED, "comm_unique_id: call maed_comm_load first");
    MAED_CHECK_ARG(id128, MAED_ERR_ARG, "comm_unique_id: null pointer");
    static_assert(sizeof(ncclUniqueId) == MAED_COMM_ID_BYTES, "ncclUniqueId size");
    COMM_NCCL(g_api.GetUniqueId((ncclUniqueId*)id128), "ncclGetUniqueId");
    return MAED_OK;
}

extern "C" int maed_comm_init(int rank, int world, const void* unique_id) {
    MAED_CHECK_ARG(g_api.handle, MAED_ERR_UNSUPPORTED, "comm_init: call maed_comm_load first");
    MAED_CHECK_ARG(!g_comm, MAED_ERR_ARG, "comm_init: communicator already initialised");
    MAED_CHECK_ARG(unique_id && world > 0 && rank >= 0 && rank < world, MAED_ERR_ARG, "comm_init: bad rank %d / world %d", rank, world);
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    COMM_HIP(hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking), "comm_init: side stream");
    COMM_HIP(hipEventCreateWithFlags(&g_ev_ready, hipEventDisableTiming), "comm_init: event");
    COMM_HIP(hipEventCreateWithFlags(&g_ev_done, hipEventDisableTiming), "comm_init: event");
    COMM_NCCL(g_api.CommInitRank(&g_comm, world, id, rank), "ncclCommInitRank");
    g_world = world;
    return MAED_OK;
}

extern "C" int maed_comm_allreduce_async(void* buf, size_t n, int dtype, void* compute_stream) {
    MAED_CHECK_ARG(g_comm, MAED_ERR_UNSUPPORTED, "comm_allreduce: communicator not initialised");
    MAED_CHECK_ARG(buf || n == 0, MAED_ERR_ARG, "comm_allreduce: null buffer");
    MAED_CHECK_ARG(dtype == MAED_F32 || dtype == MAED_BF16, MAED_ERR_ARG, "comm_allreduce: bad dtype %d", dtype);
    if (n == 0) return MAED_OK;
    COMM_HIP(hipEventRecord(g_ev_ready, (hipStream_t)compute_stream), "comm_allreduce: record");
    COMM_HIP(hipStreamWaitEvent(g_side, g_ev_ready, 0), "comm_allreduce: wait");
    COMM_NCCL(g_api.AllReduce(buf, buf, n, dtype == MAED_F32 ? ncclFloat32 : ncclBfloat16, ncclSum, g_comm, g_side), "ncclAllReduce");
    return MAED_OK;
}

extern "C" int maed_comm_wait(void* compute_stream) {
    MAED_CHECK_ARG(g_comm, MAED_ERR_UNSUPPORTED, "comm_wait: communicator not initialised");
    COMM_HIP(hipEventRecord(g_ev_done, g_side), "comm_wait: record");
    COMM_HIP(hipStreamWaitEvent((hipStream_t)compute_stream, g_ev_done, 0), "comm_wait: wait");
    return MAED_OK;
}

extern "C" int maed_comm_world(void) { return g_comm ? g_world : 0; }

extern "C" int maed_comm_destroy(void) {
    if (g_comm) {
        COMM_HIP(hipStreamSynchronize(g_side), "comm_destroy: sync");
        COMM_NCCL(g_api.CommDestroy(g_comm), "ncclCommDestroy");
        g_comm = nullptr;
    }
    if (g_ev_ready) { (void)hipEventDestroy(g_ev_ready); g_ev_ready = nullptr; }
    if (g_ev_done) { (void)hipEventDestroy(g_ev_done); g_ev_done = nullptr; }
    if (g_side) { (void)hipStreamDestroy(g_side); g_side = nullptr; }
    g_world = 0;
    return MAED_OK;
}
