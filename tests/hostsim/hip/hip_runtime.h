// TEST INFRASTRUCTURE -- host simulation of the small subset of the HIP programming model that the
// decoder-tail / loss kernels of libmaed_hip use, so their ARITHMETIC can be checked on a machine without a
// GPU (tests/test_hostsim_*.py).  It is found as <hip/hip_runtime.h> when the kernel sources are compiled
// for x86 by tests/hostsim/build_sim.py; nothing in maed_amd/ ever loads the resulting library.
//
// Model: every GPU thread of a workgroup is a real host thread; workgroups run one after another.
//   __syncthreads()      -> block barrier;  __shfl_xor / MFMA -> exchange through a per-wave (64 threads) barrier
//   __shared__           -> function-local static (one workgroup is alive at a time)
//   atomicAdd            -> std::atomic_ref
// Threads that return early drop out of the barriers, as exited lanes do on hardware.
#pragma once
#define MAED_HOSTSIM 1
#include <algorithm>
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __constant__ const
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 { float x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hostsim"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }

namespace hostsim {
struct WaveCtx {
    std::barrier<> bar;
    // every exchange area exists twice: lane-exchange k uses copy k & 1 and needs ONE barrier (write, barrier, read) -- a lane can be at
    // most one exchange ahead of the slowest lane of its wave (it cannot pass barrier k+1 before everybody arrived there, i.e. finished
    // reading copy k & 1), and exchange k+1 writes the other copy
    float fx[2][64];
    float fa[2][64], fb[2][64];
    float ha[2][64][8], hb[2][64][8];   // bf16 MFMA operands, widened
    uint64_t u64[2][64];                // readfirstlane / LDS-DMA base exchange
    int live;                       // threads of this wave that exist (the last wave of a block may be partial)
    explicit WaveCtx(int n) : bar(n), live(n) {}
};
struct BlockCtx {
    std::barrier<> bar;
    std::vector<std::unique_ptr<WaveCtx>> waves;
    explicit BlockCtx(int n) : bar(n) {
        for (int w = 0; w * 64 < n; ++w) waves.emplace_back(new WaveCtx(std::min(64, n - w * 64)));
    }
};
struct Idx { unsigned x, y, z; };
extern thread_local Idx t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
extern thread_local BlockCtx* t_block;
extern thread_local int t_tid;
extern thread_local unsigned t_wop;     // wave-exchange counter of this lane (same sequence in every lane of a wave)
extern thread_local void* t_dyn_lds;

template <typename K, typename... Args>
void launch(K kernel, dim3 grid, dim3 block, size_t dyn_lds, hipStream_t, Args... args) {
    // one host thread per GPU thread of a workgroup, created ONCE per launch; the threads walk the workgroups of the grid one after
    // another (function-local `static` LDS arrays mean only one workgroup may be alive at a time), separated by a full barrier
    const int nthr = (int)(block.x * block.y * block.z);
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    if (nblocks == 0 || nthr == 0) return;
    std::vector<double> dyn((dyn_lds + 7) / 8 + 2);   // dynamic LDS of the workgroup (16-B aligned)
    std::unique_ptr<BlockCtx> ctx;
    // ONE full barrier per workgroup boundary; its completion step (run by the last thread to arrive, while the others are parked) gives
    // the next workgroup fresh block / wave barriers (exited lanes drop out of them) and poisons the dynamic LDS region: LDS content is
    // undefined at workgroup start on hardware, so a kernel reading LDS it never wrote fails its parity test here instead of passing on zeros
    auto next_block = [&]() noexcept {
        ctx.reset(new BlockCtx(nthr));
        memset(dyn.data(), 0xFF, dyn.size() * sizeof(double));
    };
    next_block();
    std::barrier<decltype(next_block)> between(nthr, next_block);
    std::vector<std::thread> th;
    th.reserve(nthr);
    for (int t = 0; t < nthr; ++t)
        th.emplace_back([&, t]() {
            t_threadIdx = Idx{(unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y)};
            t_blockDim = Idx{block.x, block.y, block.z};
            t_gridDim = Idx{grid.x, grid.y, grid.z};
            t_tid = t;
            t_dyn_lds = (void*)(((uintptr_t)dyn.data() + 15) & ~(uintptr_t)15);
            for (size_t b = 0; b < nblocks; ++b) {
                t_blockIdx = Idx{(unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((size_t)grid.x * grid.y))};
                t_block = ctx.get();
                t_wop = 0;
                kernel(args...);
                ctx->waves[t / 64]->bar.arrive_and_drop();
                ctx->bar.arrive_and_drop();
                between.arrive_and_wait();                          // nobody still uses this workgroup's context / LDS; completion sets up the next
            }
        });
    for (auto& t : th) t.join();
}
}  // namespace hostsim

#define MAED_DYN_SHARED(T, name) T* name = (T*)hostsim::t_dyn_lds
#define threadIdx hostsim::t_threadIdx
#define blockIdx hostsim::t_blockIdx
#define blockDim hostsim::t_blockDim
#define gridDim hostsim::t_gridDim
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) hostsim::launch(kernel, grid, block, lds, stream, __VA_ARGS__)

static inline void __syncthreads() { hostsim::t_block->bar.arrive_and_wait(); }

// lanes of one wave exchanging data through LDS without a block barrier (hardware: the wave's LDS operations execute in order): here the lanes are host
// threads, so the hand-over needs the wave's barrier
static inline void hostsim_wave_lds_sync() { hostsim::t_block->waves[hostsim::t_tid / 64]->bar.arrive_and_wait(); }

static inline float __shfl_xor(float v, int mask, int /*width*/ = 64) {
    hostsim::WaveCtx& w = *hostsim::t_block->waves[hostsim::t_tid / 64];
    const int lane = hostsim::t_tid % 64;
    const unsigned p = hostsim::t_wop++ & 1u;
    w.fx[p][lane] = v;
    w.bar.arrive_and_wait();
    return w.fx[p][lane ^ mask];
}

static inline int __shfl_xor(int v, int mask, int width = 64) {
    return __builtin_bit_cast(int, __shfl_xor(__builtin_bit_cast(float, v), mask, width));
}
static inline float __shfl(float v, int src_lane, int /*width*/ = 64) {
    hostsim::WaveCtx& w = *hostsim::t_block->waves[hostsim::t_tid / 64];
    const int lane = hostsim::t_tid % 64;
    const unsigned p = hostsim::t_wop++ & 1u;
    w.fx[p][lane] = v;
    w.bar.arrive_and_wait();
    return w.fx[p][src_lane & 63];
}
// wave vote: true if the predicate holds in any live lane
static inline int __any(int pred) {
    hostsim::WaveCtx& w = *hostsim::t_block->waves[hostsim::t_tid / 64];
    const int lane = hostsim::t_tid % 64;
    const unsigned p = hostsim::t_wop++ & 1u;
    w.fx[p][lane] = pred ? 1.0f : 0.0f;
    w.bar.arrive_and_wait();
    int r = 0;
    for (int l = 0; l < w.live; ++l) r |= (w.fx[p][l] != 0.0f);
    return r;
}
static inline int __double2loint(double d) { return (int)(uint32_t)(__builtin_bit_cast(uint64_t, d) & 0xffffffffu); }
static inline int __double2hiint(double d) { return (int)(uint32_t)(__builtin_bit_cast(uint64_t, d) >> 32); }
static inline double __hiloint2double(int hi, int lo) {
    return __builtin_bit_cast(double, ((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo);
}

typedef float hostsim_f32x16 __attribute__((ext_vector_type(16)));
// v_mfma_f32_32x32x2_f32: A[i][k] from lane k*32+i, B[k][j] from lane k*32+j, D[(r&3)+8*(r>>2)+4*(lane>>5)][lane&31] in reg r
static inline hostsim_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, hostsim_f32x16 acc, int, int, int) {
    hostsim::WaveCtx& w = *hostsim::t_block->waves[hostsim::t_tid / 64];
    const int lane = hostsim::t_tid % 64;
    const unsigned p = hostsim::t_wop++ & 1u;
    w.fa[p][lane] = a; w.fb[p][lane] = b;
    w.bar.arrive_and_wait();
    const int j = lane & 31, hi = lane >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        acc[r] = fmaf(w.fa[p][32 + i], w.fb[p][32 + j], fmaf(w.fa[p][i], w.fb[p][j], acc[r]));
    }
    return acc;
}

typedef __bf16 hostsim_bf16x8 __attribute__((ext_vector_type(8)));
// v_mfma_f32_32x32x16_bf16: A[i][k] = lane (k/8)*32 + i, element k%8;  B[k][j] = lane (k/8)*32 + j, element k%8;  D as above
static inline hostsim_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(hostsim_bf16x8 a, hostsim_bf16x8 b, hostsim_f32x16 acc, int, int, int) {
    hostsim::WaveCtx& w = *hostsim::t_block->waves[hostsim::t_tid / 64];
    const int lane = hostsim::t_tid % 64;
    const unsigned p = hostsim::t_wop++ & 1u;
    for (int e = 0; e < 8; ++e) { w.ha[p][lane][e] = (float)a[e]; w.hb[p][lane][e] = (float)b[e]; }
    w.bar.arrive_and_wait();
    const int j = lane & 31, hi = lane >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float s = acc[r];
        for (int kg = 0; kg < 2; ++kg)
            for (int e = 0; e < 8; ++e) s = fmaf(w.ha[p][kg * 32 + i][e], w.hb[p][kg * 32 + j][e], s);
        acc[r] = s;
    }
    return acc;
}

// wave-uniform value of the first live lane
static inline int __builtin_amdgcn_readfirstlane(int v) {
    hostsim::WaveCtx& w = *hostsim::t_block->waves[hostsim::t_tid / 64];
    const int lane = hostsim::t_tid % 64;
    const unsigned p = hostsim::t_wop++ & 1u;
    if (lane == 0) w.u64[p][0] = (uint64_t)(uint32_t)v;
    w.bar.arrive_and_wait();
    return (int)(uint32_t)w.u64[p][0];
}
// global_load_lds_dwordx4: every lane's 16 bytes land at (lane 0's LDS pointer) + lane * size
static inline void __builtin_amdgcn_global_load_lds(const void* gptr, void* lds_ptr, unsigned size, int offset, unsigned) {
    hostsim::WaveCtx& w = *hostsim::t_block->waves[hostsim::t_tid / 64];
    const int lane = hostsim::t_tid % 64;
    const unsigned p = hostsim::t_wop++ & 1u;
    if (lane == 0) w.u64[p][1] = (uint64_t)(uintptr_t)lds_ptr;
    w.bar.arrive_and_wait();
    memcpy((char*)(uintptr_t)w.u64[p][1] + (size_t)lane * size + offset, (const char*)gptr + offset, size);   // visible to readers after the kernel's own barrier
}
// v_mfma_f32_16x16x32_bf16: A[i][k] = lane (k/8)*16 + i, element k%8;  B[k][j] = lane (k/8)*16 + j, element k%8;  D[4*(lane>>4) + r][lane & 15] in reg r
typedef float hostsim_f32x4 __attribute__((ext_vector_type(4)));
static inline hostsim_f32x4 __builtin_amdgcn_mfma_f32_16x16x32_bf16(hostsim_bf16x8 a, hostsim_bf16x8 b, hostsim_f32x4 acc, int, int, int) {
    hostsim::WaveCtx& w = *hostsim::t_block->waves[hostsim::t_tid / 64];
    const int lane = hostsim::t_tid % 64;
    const unsigned p = hostsim::t_wop++ & 1u;
    for (int e = 0; e < 8; ++e) { w.ha[p][lane][e] = (float)a[e]; w.hb[p][lane][e] = (float)b[e]; }
    w.bar.arrive_and_wait();
    const int j = lane & 15, q = lane >> 4;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * q + r;
        float s = acc[r];
        for (int kg = 0; kg < 4; ++kg)
            for (int e = 0; e < 8; ++e) s = fmaf(w.ha[p][kg * 16 + i][e], w.hb[p][kg * 16 + j][e], s);
        acc[r] = s;
    }
    return acc;
}
// v_perm_b32: byte select from the 8-byte pool {src0 (bytes 7..4), src1 (bytes 3..0)}; selector values 0..7 only (what the kernels use)
static inline uint32_t __builtin_amdgcn_perm(uint32_t src0, uint32_t src1, uint32_t sel) {
    const uint64_t pool = ((uint64_t)src0 << 32) | src1;
    uint32_t r = 0;
    for (int b = 0; b < 4; ++b) r |= (uint32_t)((pool >> (8 * ((sel >> (8 * b)) & 7))) & 0xff) << (8 * b);
    return r;
}
// ds_read_b64_tr_b16: every lane reads 4 x 16 bit at ITS OWN address; inside each 16-lane group the 16 x 4 elements are transposed:
// lane i, element j <- the element (i & 3) that lane 4j + (i >> 2) of the group loaded (column i of a 4 x 16 block whose row j is
// covered by lanes 4j .. 4j+3)
typedef short hostsim_v4i16 __attribute__((ext_vector_type(4)));
static inline hostsim_v4i16 __builtin_amdgcn_ds_read_tr16_b64_v4i16(hostsim_v4i16* p) {
    hostsim::WaveCtx& w = *hostsim::t_block->waves[hostsim::t_tid / 64];
    const int lane = hostsim::t_tid % 64;
    const unsigned q = hostsim::t_wop++ & 1u;
    uint64_t mine;
    memcpy(&mine, p, 8);
    w.u64[q][lane] = mine;
    w.bar.arrive_and_wait();
    hostsim_v4i16 r;
    const int i = lane & 15, g = lane & ~15;
    for (int j = 0; j < 4; ++j) {
        const uint64_t src = w.u64[q][g + 4 * j + (i >> 2)];
        r[j] = (short)(uint16_t)(src >> (16 * (i & 3)));
    }
    return r;
}
static inline void __builtin_amdgcn_s_barrier() { hostsim::t_block->bar.arrive_and_wait(); }
static inline void __builtin_amdgcn_s_setprio(int) {}          // scheduling hints: nothing to simulate
static inline void __builtin_amdgcn_sched_barrier(int) {}
typedef void* hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
// side streams / fences: everything runs synchronously on the host, so a second stream is the same stream and a fence is nothing
#define hipStreamNonBlocking 1u
#define hipEventDisableTiming 2u
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
#define hipFuncAttributeMaxDynamicSharedMemorySize 0
template <typename K> static inline hipError_t hipFuncSetAttribute(K, int, int) { return hipSuccess; }

template <typename T> static inline T atomicAdd(T* p, T v) { return std::atomic_ref<T>(*p).fetch_add(v); }

static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline int64_t min(int64_t a, int64_t b) { return a < b ? a : b; }
static inline int64_t max(int64_t a, int64_t b) { return a > b ? a : b; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
#define __expf(x) expf(x)
#define __logf(x) logf(x)
#define __log2f(x) log2f(x)
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
