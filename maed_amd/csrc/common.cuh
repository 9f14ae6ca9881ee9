// Shared device helpers for libmaed_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include "../../include/maed_hip.h"

#define HEAD_DIM 64

// the two places where kernels speak raw ISA; tests/hostsim (x86 build of the same sources) substitutes no-ops / plain pointers
#ifdef MAED_HOSTSIM
#define MAED_WAIT_VMCNT0() do { } while (0)
#define MAED_WAIT_VMCNT(n) do { } while (0)
#define MAED_WAIT_LGKMCNT0() do { } while (0)
#define MAED_LDS_DMA16(base_, voff_, lds_ptr_) __builtin_amdgcn_global_load_lds((const char*)(base_) + (voff_), (void*)(lds_ptr_), 16, 0, 0)
#define MAED_LDS_DMA16_PTR(gptr_, lds_ptr_) __builtin_amdgcn_global_load_lds((const void*)(gptr_), (void*)(lds_ptr_), 16, 0, 0)
typedef void maed_lds_void_t;
typedef const void maed_glb_void_t;
#define MAED_DS_READ_TR16(p_) __builtin_amdgcn_ds_read_tr16_b64_v4i16((hostsim_v4i16*)(p_))
#define MAED_WAVE_LDS_SYNC() hostsim_wave_lds_sync()
#else
// between LDS writes of one wave and reads of the same data by OTHER lanes of that wave: the hardware executes a wave's LDS operations in order, so only the
// compiler has to keep them in program order (the host simulator runs lanes as threads and needs a real wave barrier here)
#define MAED_WAVE_LDS_SYNC() __builtin_amdgcn_wave_barrier()
#define MAED_WAIT_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define MAED_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")      // counted: the n most recent VMEM operations stay in flight
#define MAED_WAIT_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// one LDS-DMA of 16 B per lane in its cheapest form: wave-uniform 64-bit base in SGPRs + 32-bit lane offset, LDS destination = the
// wave-uniform byte address in M0 (+ lane * 16).  Through the builtin hipcc keeps 64-bit lane pointers (two v_lshl_add_u64 per DMA)
// and may route the LDS base through v_readfirstlane.  M0 is written in the statement that reads it (it is compiler-reserved and not
// preserved across statements); hipcc does not count this load: every consumer waits with MAED_WAIT_VMCNT + a barrier.
#define MAED_LDS_DMA16(base_, voff_, lds_ptr_)                                                                                   \
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2"                                                 \
                 :: "v"((uint32_t)(voff_)), "s"((uint32_t)(uintptr_t)(lds_ptr_)), "s"((const char*)(base_)) : "memory")
// the same with a 64-bit source pointer PER LANE (gathers whose lanes point into different allocations: a tensor and the page of zeros): two VGPRs of address per copy,
// otherwise as above -- uncounted by hipcc, so a ring of several tiles stays in flight across __syncthreads() (the builtin's copies are pending LDS writes to the
// compiler: its barrier fence drains them with vmcnt(0))
#define MAED_LDS_DMA16_PTR(gptr_, lds_ptr_)                                                                                      \
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"                                               \
                 :: "v"((const char*)(gptr_)), "s"((uint32_t)(uintptr_t)(lds_ptr_)) : "memory")
typedef __attribute__((address_space(3))) void maed_lds_void_t;
typedef const __attribute__((address_space(1))) void maed_glb_void_t;
// ds_read_b64_tr_b16: every lane reads 8 bytes at its own LDS address; inside each 16-lane group the 16 x 4 elements come back transposed
typedef short maed_v4i16_t __attribute__((ext_vector_type(4)));
#define MAED_DS_READ_TR16(p_) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) maed_v4i16_t*)(p_))
#endif

// ---- buffer loads: 16 bytes per lane at (wave-uniform 128-bit resource: base + extent) + 32-bit lane byte offset + wave-uniform byte offset.  No 64-bit lane
// arithmetic per load (hipcc keeps 64-bit lane pointers for global_load: two VALU instructions each), and the hardware range-checks the LANE offset against the
// extent: a lane whose offset lies past it reads zeros -- rows past the end of a tensor and out-of-image convolution taps (lane offset = MAED_BUF_OOB) cost no
// select and no branch.  (The uniform offset is NOT part of the check: callers keep it inside the tensor.)
#define MAED_BUF_OOB 0xfffffff0u
#ifdef MAED_HOSTSIM
struct maed_buf_t { const char* base; uint32_t bytes; };
static inline maed_buf_t maed_make_buf(const void* p, uint64_t bytes) { return maed_buf_t{(const char*)p, (uint32_t)(bytes > 0xffffffffull ? 0xffffffffull : bytes)}; }
static inline uint4 maed_buf_load16(const maed_buf_t& r, uint32_t lane_off, uint32_t uniform_off) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if ((uint64_t)lane_off + 16 <= r.bytes) memcpy(&v, r.base + lane_off + uniform_off, 16);
    return v;
}
#else
typedef __amdgpu_buffer_rsrc_t maed_buf_t;
typedef unsigned int maed_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ maed_buf_t maed_make_buf(const void* p, uint64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(uint32_t)(bytes > 0xffffffffull ? 0xffffffffull : bytes), 0x00020000);
}
__device__ __forceinline__ uint4 maed_buf_load16(maed_buf_t r, uint32_t lane_off, uint32_t uniform_off) {
    const maed_u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off, (int)uniform_off, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
#endif
__device__ __forceinline__ float4 maed_buf_load_f4(const maed_buf_t& r, uint32_t lane_off, uint32_t uniform_off) {
    const uint4 v = maed_buf_load16(r, lane_off, uniform_off);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// dynamic LDS of a kernel (the host simulator of tests/hostsim substitutes its own definition)
#ifndef MAED_DYN_SHARED
#define MAED_DYN_SHARED(T, name) extern __shared__ __attribute__((aligned(16))) T name[]
#endif

struct bf16 { unsigned short v; };

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// ---- error plumbing (host) -------------------------------------------------------------------
void maed_set_error(const char* fmt, ...);
#define MAED_CHECK_ARG(cond, code, ...) do { if (!(cond)) { maed_set_error(__VA_ARGS__); return (code); } } while (0)
#define MAED_CHECK_LAUNCH(name) do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) { \
    maed_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); return MAED_ERR_LAUNCH; } } while (0)
#define MAED_PROPAGATE(expr) do { int rc__ = (expr); if (rc__ != MAED_OK) return rc__; } while (0)
// a HIP runtime call inside an entry point (memset, event, stream wait): its failure is the entry point's failure
#define MAED_HIP(expr, name) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { \
    maed_set_error("%s: %s", name, hipGetErrorString(e__)); return MAED_ERR_LAUNCH; } } while (0)

// the streams and event rings the library owns: all created by this one function (block.hip), which maed_init calls
int maed_init_runtime(void);
// process-wide options (csrc/options.hip; maed_set_option in include/maed_hip.h)
int maed_opt(int key);

static inline bool is_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// ---- scalar conversions ----------------------------------------------------------------------
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((uint32_t)h) << 16); }
typedef __bf16 bf16x2_hw_t __attribute__((ext_vector_type(2)));
// fp32 -> bf16 round-to-nearest-even on the gfx950 converter (v_cvt_pk_bf16_f32): one instruction per PAIR
__device__ __forceinline__ unsigned short f2bf(float f) { const __bf16 h = (__bf16)f; return __builtin_bit_cast(unsigned short, h); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const bf16x2_hw_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

// c + a.lo * b.lo + a.hi * b.hi on packed bf16 pairs (v_dot2c_f32_bf16): sums / sums of squares of ROUNDED outputs without unpacking them
#ifdef MAED_HOSTSIM
static inline float maed_dot2_bf16(uint32_t a, uint32_t b, float c) {
    return c + (bf2f((unsigned short)(a & 0xffffu)) * bf2f((unsigned short)(b & 0xffffu)) + bf2f((unsigned short)(a >> 16)) * bf2f((unsigned short)(b >> 16)));
}
#else
__device__ __forceinline__ float maed_dot2_bf16(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_hw_t, a), __builtin_bit_cast(bf16x2_hw_t, b), c, false);
}
#endif

__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16* p) { return bf2f(p->v); }
__device__ __forceinline__ void stf(float* p, float x) { *p = x; }
__device__ __forceinline__ void stf(bf16* p, float x) { p->v = f2bf(x); }

// 4-element vector access (16 B for float, 8 B for bf16); pointers must be 4-element aligned
__device__ __forceinline__ void ld4(const float* p, float (&o)[4]) {
    float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void ld4(const bf16* p, float (&o)[4]) {
    uint2 v = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}
__device__ __forceinline__ void st4(float* p, const float (&o)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
}
__device__ __forceinline__ void st4(bf16* p, const float (&o)[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
}
// 8 contiguous elements
__device__ __forceinline__ void ld8(const float* p, float (&o)[8]) {
    float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void ld8(const bf16* p, float (&o)[8]) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
    o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ void st8(float* p, const float (&o)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
}
__device__ __forceinline__ void st8(bf16* p, const float (&o)[8]) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]),
                                             pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7]));
}

// 8 bf16 with a non-temporal store: data nobody reads again in this pass (the bf16 twins an fp32 forward leaves for the backward) should not displace the forward's
// working set from L2 / the Infinity Cache
__device__ __forceinline__ void st8_nt(bf16* p, const float (&o)[8]) {
#ifdef MAED_HOSTSIM
    st8(p, o);
#else
    typedef uint32_t maed_u32x4_nt_t __attribute__((ext_vector_type(4)));
    const maed_u32x4_nt_t v = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7])};
    __builtin_nontemporal_store(v, reinterpret_cast<maed_u32x4_nt_t*>(p));
#endif
}

// ---- wave (64 lanes) reductions -----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// dst[c] += sum_r src[r][c] for a small (rows x cols) fp32 matrix of partial sums (closing step of the deferred dgamma/dbeta paths).
// 64 columns x 4 row lanes per workgroup, <= 64 rows per workgroup (grid.y row chunks): at most 16 independent, coalesced loads per
// thread, one LDS fold, one atomic per column and row chunk -- a handful per address, unlike the per-workgroup atomics it replaces.
// dst_of(c) maps a column to its destination (dgamma / dbeta are separate or interleaved depending on the producer).
template <typename DstOf>
__device__ __forceinline__ void colsum_add(const float* __restrict__ src, int rows, int cols, DstOf dst_of) {
    __shared__ float fold[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int r0 = blockIdx.y * 64, r1 = min(rows, r0 + 64);
    float t = 0.f;
    if (c < cols)
        for (int r = r0 + rl; r < r1; r += 4) t += src[(int64_t)r * cols + c];
    fold[rl][threadIdx.x & 63] = t;
    __syncthreads();
    if (rl == 0 && c < cols) atomicAdd(dst_of(c), (fold[0][threadIdx.x] + fold[1][threadIdx.x]) + (fold[2][threadIdx.x] + fold[3][threadIdx.x]));
}

// ---- hand-offs between workgroups INSIDE a launch (the one-pass GroupNorm backward, the fused attentive addition) ---------------------------------------
// gfx950: eight XCDs with private L2s, a CU's vector L1 is never refreshed by another CU's stores.  The forms used here (MI355X_MICROARCH.md, "inter-workgroup
// visibility"): payload as device-scope atomics (adds, or relaxed agent stores = write-through `sc1` stores) -> every storing wave drains (vmcnt 0) -> barrier ->
// ONE lane adds to the arrival counter and polls it relaxed; consumers read the payload with device-scope loads / returning atomics.  No L2 write-back, no L1
// invalidate.  The spin is bounded: a peer that never arrives (dirty counter, lost launch) turns into a poisoned result, not a hung GPU.
// The host simulator runs workgroups one after another and never reaches the spin (the callers split such kernels into two launches there).
#ifdef MAED_HOSTSIM
__device__ __forceinline__ bool maed_frame_arrive_and_wait(uint32_t* ctr, uint32_t S) { atomicAdd(ctr, 1u); return *ctr >= S; }
__device__ __forceinline__ float maed_coherent_read(float* p) { return *p; }
__device__ __forceinline__ void maed_agent_store(float* p, float v) { *p = v; }
__device__ __forceinline__ float maed_agent_load(const float* p) { return *p; }
#else
__device__ __forceinline__ bool maed_frame_arrive_and_wait(uint32_t* ctr, uint32_t S) {      // ONE lane; every wave has drained its payload (vmcnt 0 + barrier) before
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (!= and not <: exactly S workgroups arrive at a counter that was zero at launch -- a counter that was NOT cleared can then never release the barrier early
    //  with stale data behind it; it runs into the spin bound and poisons the result: loud instead of subtly wrong)
    for (uint32_t spins = 0; __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != S; ) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1u << 21)) return false;
    }
    return true;
}
__device__ __forceinline__ float maed_coherent_read(float* p) { return __hip_atomic_fetch_add(p, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void maed_agent_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float maed_agent_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif

// A bounded spin that expired (the workgroups of a frame were not co-resident: GPU shared with another process, preemption, a debugger) poisons THAT call's result
// with NaN -- and tells the host: one system-scope add to the library's fault word (pinned host memory, options.hip).  The launchers of the kernels that use the
// barrier look at the word before every launch: once it is non-zero they take their two-pass forms for the rest of the process, maed_device_faults() returns the
// count and maed_last_error() says what happened.
uint32_t* maed_fault_word(void);
bool maed_fault_seen(const char* who);
__device__ __forceinline__ void maed_report_fault(uint32_t* w) {
#ifdef MAED_HOSTSIM
    if (w) ++*w;
#else
    if (w) __hip_atomic_fetch_add(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}

// ---- math ------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float dgelu_erf(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// bf16 throughput mode: erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below bf16 resolution) -- one v_exp,
// one v_rcp and a 5-term Horner chain instead of libm erff (~3x fewer VALU issues in the GELU GEMM epilogues).
// exp(-x^2/2) is shared between the erf tail and the Gaussian pdf of the derivative.
__device__ __forceinline__ void gelu_parts_fast(float x, float& cdf, float& pdf) {
    const float ax = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    const float e = __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);   // exp(-x^2/2)
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float erf_abs = 1.0f - poly * t * e;
    cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
    pdf = 0.39894228040143267794f * e;
}
template <typename T> __device__ __forceinline__ float gelu_fwd(float x);
template <> __device__ __forceinline__ float gelu_fwd<float>(float x) { return gelu_erf(x); }      // parity mode: libm erff
template <> __device__ __forceinline__ float gelu_fwd<struct bf16>(float x) { float c, p; gelu_parts_fast(x, c, p); return x * c; }
template <typename T> __device__ __forceinline__ float gelu_bwd(float x);
template <> __device__ __forceinline__ float gelu_bwd<float>(float x) { return dgelu_erf(x); }
template <> __device__ __forceinline__ float gelu_bwd<struct bf16>(float x) { float c, p; gelu_parts_fast(x, c, p); return fmaf(x, p, c); }

// value a float takes after a round trip through storage type T (what a later kernel reading the stored tensor sees)
template <typename T> __device__ __forceinline__ float round_to(float x);
template <> __device__ __forceinline__ float round_to<float>(float x) { return x; }
template <> __device__ __forceinline__ float round_to<struct bf16>(float x) { return bf2f(f2bf(x)); }

template <typename T> struct dtype_of;
template <> struct dtype_of<float> { static constexpr int value = MAED_F32; };
template <> struct dtype_of<bf16> { static constexpr int value = MAED_BF16; };

static inline size_t dtype_size(int dtype) { return dtype == MAED_BF16 ? 2 : 4; }

// dispatch a templated launcher on the runtime dtype
#define MAED_DISPATCH_DTYPE(dtype, T, ...) do { \
    if ((dtype) == MAED_F32) { using T = float; __VA_ARGS__; } \
    else if ((dtype) == MAED_BF16) { using T = bf16; __VA_ARGS__; } \
    else { maed_set_error("bad dtype %d", (int)(dtype)); return MAED_ERR_ARG; } } while (0)
