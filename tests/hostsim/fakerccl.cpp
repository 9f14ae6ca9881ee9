// TEST INFRASTRUCTURE -- libfakerccl.so: the five NCCL entry points csrc/comm.hip binds with dlsym (ncclGetUniqueId, ncclCommInitRank, ncclAllReduce,
// ncclCommDestroy, ncclGetErrorString), implemented over POSIX shared memory for RANKS THAT ARE PROCESSES OF ONE HOST and buffers that are host memory (the host
// simulator's "device" pointers).  Purpose: drive maed_comm_load / unique_id / init / allreduce_async / wait / destroy from two processes on a
// box without GPUs -- RCCL itself refuses two ranks on one device and cannot run on none.  Never loaded by the product: maed_comm_load takes whatever path the
// host names, and only tests/test_comm_world2.py names this one.
//
// Protocol: the unique id is the name of a shared-memory segment; every rank maps it, copies its chunk into slot[rank], meets at a sense-reversing barrier,
// sums the slots IN RANK ORDER (every rank computes bit-identical results, as a ring all-reduce does) into its receive buffer, meets again.  Synchronous: the call
// returns when the reduction is done (the simulator's streams are synchronous too).
#include <atomic>
#include <fcntl.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include "rccl/rccl.h"

namespace {
constexpr int kMaxRanks = 8;
constexpr size_t kChunkBytes = 1 << 20;
struct Shared {
    std::atomic<int> arrived, generation, attached, detached;
    alignas(64) unsigned char slot[kMaxRanks][kChunkBytes];
};
}  // namespace
struct ncclComm { Shared* sh; int rank, world; char name[64]; };

static const char* g_last = "ok";

static bool barrier(ncclComm* c) {
    Shared* s = c->sh;
    const int gen = s->generation.load(std::memory_order_acquire);
    if (s->arrived.fetch_add(1, std::memory_order_acq_rel) == c->world - 1) {
        s->arrived.store(0, std::memory_order_relaxed);
        s->generation.store(gen + 1, std::memory_order_release);
        return true;
    }
    for (long spins = 0; s->generation.load(std::memory_order_acquire) == gen; ++spins) {
        sched_yield();
        if (spins > 200000000L) { g_last = "fakerccl: a rank never arrived at the barrier"; return false; }
    }
    return true;
}

static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof id->internal, "/maedfakerccl_%d_%lx%lx", (int)getpid(), (long)ts.tv_sec, (long)ts.tv_nsec);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks || id.internal[0] != '/') { g_last = "fakerccl: bad arguments"; return ncclInvalidArgument; }
    const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { g_last = "fakerccl: shm_open failed"; return ncclSystemError; }
    if (ftruncate(fd, sizeof(Shared)) != 0) { close(fd); g_last = "fakerccl: ftruncate failed"; return ncclSystemError; }       // (a fresh segment is zero-filled: counters start at 0)
    void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { g_last = "fakerccl: mmap failed"; return ncclSystemError; }
    ncclComm* c = new ncclComm{(Shared*)p, rank, nranks, {0}};
    strncpy(c->name, id.internal, sizeof c->name - 1);
    c->sh->attached.fetch_add(1);
    for (long spins = 0; c->sh->attached.load() < nranks; ++spins) {          // like ncclCommInitRank: returns when every rank has joined
        sched_yield();
        if (spins > 200000000L) { g_last = "fakerccl: not every rank called ncclCommInitRank"; return ncclSystemError; }
    }
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c, void* /*stream*/) {
    if (!c || op != ncclSum || (dt != ncclFloat32 && dt != ncclBfloat16)) { g_last = "fakerccl: only sum of float32 / bfloat16"; return ncclInvalidArgument; }
    const size_t es = dt == ncclFloat32 ? 4 : 2, per = kChunkBytes / es;
    for (size_t o = 0; o < count; o += per) {
        const size_t n = count - o < per ? count - o : per;
        memcpy(c->sh->slot[c->rank], (const char*)sendbuff + o * es, n * es);
        if (!barrier(c)) return ncclSystemError;
        if (dt == ncclFloat32) {
            float* out = (float*)recvbuff + o;
            for (size_t i = 0; i < n; ++i) { float s = 0.f; for (int r = 0; r < c->world; ++r) s += ((const float*)c->sh->slot[r])[i]; out[i] = s; }
        } else {
            uint16_t* out = (uint16_t*)recvbuff + o;
            for (size_t i = 0; i < n; ++i) { float s = 0.f; for (int r = 0; r < c->world; ++r) s += bf2f(((const uint16_t*)c->sh->slot[r])[i]); out[i] = f2bf(s); }
        }
        if (!barrier(c)) return ncclSystemError;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    if (c->sh->detached.fetch_add(1) == c->world - 1) shm_unlink(c->name);       // the last rank out removes the segment
    munmap(c->sh, sizeof(Shared));
    delete c;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : g_last; }

}  // extern "C"
