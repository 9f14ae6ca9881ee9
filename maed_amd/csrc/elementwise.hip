// HBM-bound glue of the STE: attentive addition (K5), embeddings (K8), the transpose/cast feeding the
// weight-gradient GEMMs, and the fused Adam step.  All kernels use 8/16-byte accesses and fp32 math.
#include "common.cuh"

// ---------------------------------------------------------------------------------------------------
// K5 attentive addition (vision_transformer.py:152-158)
// ---------------------------------------------------------------------------------------------------
// means[f][c] = mean_p x_s[f][p][c], means[f][C + c] = mean_p x_t[f][p][c]; thread per 4 channels.
// grid (ceil(C/4/64), F, 2); a wave reads 512/1024 contiguous bytes of a row per step.
#define CM_SPLIT 8  // token-axis split per (f, channel group) so the grid fills the chip
template <typename T>
__global__ __launch_bounds__(64) void st_colmean_kernel(const T* __restrict__ x_s, const T* __restrict__ x_t, float* __restrict__ part,
                                                        int P, int C) {
    const int c4 = blockIdx.x * 64 + threadIdx.x;
    if (c4 * 4 >= C) return;
    const int f = blockIdx.y, which = blockIdx.z / CM_SPLIT, sp = blockIdx.z % CM_SPLIT;
    const T* src = (which ? x_t : x_s) + (int64_t)f * P * C + c4 * 4;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p = sp; p < P; p += CM_SPLIT) {
        float v[4];
        ld4(src + (int64_t)p * C, v);
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    float* dst = part + ((int64_t)f * 2 * C + which * C + c4 * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(dst + j, s[j]);
}
template <typename T>
__global__ void st_colmean_finish(const float* __restrict__ part, T* __restrict__ means, int64_t n, float invP) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) stf(means + i, part[i] * invP);
}

template <typename T>
__global__ __launch_bounds__(256) void st_mix_fwd_kernel(const T* __restrict__ x_s, const T* __restrict__ x_t, const float* __restrict__ logits,
                                                         T* __restrict__ mix, int64_t n4, int P, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int64_t e = i * 4;
    const int c = (int)(e % C); const int64_t f = e / ((int64_t)P * C);
    float lg[8], a[4], b[4], o[4];
    ld8(logits + f * 2 * C + 2 * c, lg);
    ld4(x_s + e, a); ld4(x_t + e, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a0 = 1.f / (1.f + __expf(lg[2 * j + 1] - lg[2 * j]));  // softmax over the pair, weight of x_s
        o[j] = b[j] * (1.f - a0) + a[j] * a0;
    }
    st4(mix + e, o);
}

// dlogits: thread per 4 channels of one frame; reduction over tokens split CM_SPLIT ways through atomics
template <typename T>
__global__ __launch_bounds__(64) void st_mix_bwd_reduce_kernel(const T* __restrict__ dmix, const T* __restrict__ x_s, const T* __restrict__ x_t,
                                                               float* __restrict__ part, int P, int C) {
    const int c4 = blockIdx.x * 64 + threadIdx.x;
    if (c4 * 4 >= C) return;
    const int f = blockIdx.y, sp = blockIdx.z;
    const int64_t base = (int64_t)f * P * C + c4 * 4;
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p = sp; p < P; p += CM_SPLIT) {
        float d[4], a[4], b[4];
        ld4(dmix + base + (int64_t)p * C, d); ld4(x_s + base + (int64_t)p * C, a); ld4(x_t + base + (int64_t)p * C, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) { s0[j] = fmaf(d[j], a[j], s0[j]); s1[j] = fmaf(d[j], b[j], s1[j]); }
    }
    float* dst = part + ((int64_t)f * 2 * C + 2 * c4 * 4);  // interleaved (da0, da1) per channel
#pragma unroll
    for (int j = 0; j < 4; ++j) { atomicAdd(dst + 2 * j, s0[j]); atomicAdd(dst + 2 * j + 1, s1[j]); }
}
template <typename T>
__global__ void st_mix_bwd_reduce_finish(const float* __restrict__ part, const float* __restrict__ logits, T* __restrict__ dlogits, int64_t npairs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npairs) return;
    const float l0 = logits[2 * i], l1 = logits[2 * i + 1];
    const float a0 = 1.f / (1.f + __expf(l1 - l0)), a1 = 1.f - a0;
    const float d0 = part[2 * i], d1 = part[2 * i + 1];
    const float dot = a0 * d0 + a1 * d1;
    stf(dlogits + 2 * i, a0 * (d0 - dot));
    stf(dlogits + 2 * i + 1, a1 * (d1 - dot));
}

template <typename T>
__global__ __launch_bounds__(256) void st_mix_bwd_apply_kernel(const T* __restrict__ dmix, const float* __restrict__ logits, const T* __restrict__ dmeans,
                                                               T* __restrict__ dx_s, T* __restrict__ dx_t, int64_t n4, int P, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int64_t e = i * 4;
    const int c = (int)(e % C); const int64_t f = e / ((int64_t)P * C);
    float lg[8], d[4], ms[4], mt[4], os[4], ot[4];
    ld8(logits + f * 2 * C + 2 * c, lg);
    ld4(dmix + e, d);
    ld4(dmeans + f * 2 * C + c, ms); ld4(dmeans + f * 2 * C + C + c, mt);
    const float invP = 1.f / (float)P;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a0 = 1.f / (1.f + __expf(lg[2 * j + 1] - lg[2 * j]));
        os[j] = d[j] * a0 + ms[j] * invP;
        ot[j] = d[j] * (1.f - a0) + mt[j] * invP;
    }
    st4(dx_s + e, os); st4(dx_t + e, ot);
}

// ---- K5 in ONE launch per direction (round 4, bf16) ----------------------------------------------------------------------------------------------------
// The attentive addition is small (x_s, x_t: 25 MB each at cfg3) but took four launches forward (token means, their cast, the F x 2C x 2C ts_attn GEMM, the mix)
// and four backward, two of them token-axis reductions that run at 2 TB/s.  What forces the split is one dependency: a frame's 2C logits need the token means of
// ALL its channels.  Here a frame is cut by CHANNELS: a workgroup owns 128 channels of one frame for all P tokens (2 x 50 KB of x_s / x_t at P = 197: 56 payload
// VGPRs per thread), so its token means are complete inside the workgroup; the C / 128 workgroups of a frame publish their bf16-rounded means (device-scope
// stores), meet at a per-frame counter, read the frame's 2C means back, compute THEIR 256 logits as a GEMV against the ts_attn weight (8 lanes per weight row:
// coalesced 128-byte reads, L2-resident 2 MB), and mix straight from the registers.  x_s and x_t are read once, `means` / `logits` are still written (the
// backward needs them).  Backward: the same cut -- dlogits of a channel pair are local to the workgroup, d(means) = dlogits . W_ts needs the frame's 2C dlogits.
// The arithmetic is the separate kernels' (bf16-rounded means / dlogits / dmeans where those were stored as bf16, fp32 accumulation).
// sync: F x 16 words, zero at launch (the launcher clears them); ex: F x 2C floats of exchange space.
// phase 0: everything; 1 / 2: the halves before / after the frame barrier as two launches (host simulator: workgroups run one after another).
#define ST1_NT 512
__device__ __forceinline__ void st1_unpack(const uint4& v, float (&o)[8]) {
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u); o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u); o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
}
// this workgroup's 256 GEMV outputs: out[row] = sum_k W[row_of(row)][k] * vec[k], k < K2 (K2 % 128 == 0).  8 lanes per weight row (coalesced 128-byte reads), the four
// rows a lane group owns (row = 64 q + g) advance TOGETHER, two chunks each per step: 8 independent 16-byte loads in flight per lane and K2 / 128 dependent
// steps -- the first version walked one row at a time with one load per step: 64 L2 round trips per lane, 30 us of a 47 us kernel.
template <typename RowOf, typename Emit>
__device__ __forceinline__ void st1_gemv256(const bf16* __restrict__ W, int K2, const float* vec, RowOf row_of, Emit emit) {
    const int g = threadIdx.x >> 3, l8 = threadIdx.x & 7;
    const bf16* wr[4];
    float acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { wr[q] = W + (int64_t)row_of(q * 64 + g) * K2; acc[q] = 0.f; }
    for (int k8 = l8; k8 < K2 / 8; k8 += 16) {
        uint4 w[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) { w[q][0] = *reinterpret_cast<const uint4*>(wr[q] + k8 * 8); w[q][1] = *reinterpret_cast<const uint4*>(wr[q] + (k8 + 8) * 8); }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = vec[(k8 + 8 * u) * 8 + j];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float x[8];
                st1_unpack(w[q][u], x);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[q] = fmaf(x[j], v[j], acc[q]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float a = acc[q];
        a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
        if (l8 == 0) emit(q * 64 + g, a);
    }
}

template <int CH>
__global__ __launch_bounds__(ST1_NT, 4) void st_fused_fwd_kernel(const bf16* __restrict__ x_s, const bf16* __restrict__ x_t, const bf16* __restrict__ w_ts,
                                                                 const float* __restrict__ b_ts, bf16* __restrict__ means, float* __restrict__ logits,
                                                                 bf16* __restrict__ mix, uint32_t* sync, float* ex, int P, int C, int S, uint32_t target, uint32_t* fault, int phase) {
    __shared__ float lpart[32 * 16 * 16];      // [row lane][chunk][x_s: 8 | x_t: 8] token-sum partials
    __shared__ float lvec[2048];               // the frame's 2C means
    __shared__ float llog[256];                // this slice's 128 logit pairs
    __shared__ int lfail;
    const int f = blockIdx.y, c0 = blockIdx.x * 128, tid = threadIdx.x, cb = tid & 15, rsub = tid >> 4;
    const int64_t base = (int64_t)f * P * C + c0 + cb * 8;
    uint4 xs[CH], xt[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int r = rsub + 32 * k;
        if (r < P) { xs[k] = *reinterpret_cast<const uint4*>(x_s + base + (int64_t)r * C); xt[k] = *reinterpret_cast<const uint4*>(x_t + base + (int64_t)r * C); }
        else { xs[k] = make_uint4(0u, 0u, 0u, 0u); xt[k] = xs[k]; }
    }
    if (tid == 0) lfail = 0;
    if (phase != 2) {
        float ss[8], st[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { ss[j] = 0.f; st[j] = 0.f; }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            float a[8], b[8];
            st1_unpack(xs[k], a); st1_unpack(xt[k], b);
#pragma unroll
            for (int j = 0; j < 8; ++j) { ss[j] += a[j]; st[j] += b[j]; }
        }
        float* mine = lpart + (rsub * 16 + cb) * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) { mine[j] = ss[j]; mine[8 + j] = st[j]; }
        __syncthreads();
        if (tid < 256) {
            const int which = tid >> 7, c = tid & 127;
            float t = 0.f;
            for (int r = 0; r < 32; ++r) t += lpart[(r * 16 + (c >> 3)) * 16 + which * 8 + (c & 7)];
            const unsigned short mb = f2bf(t * (1.f / (float)P));            // the bf16 the ts_attn GEMM read (means are saved for its weight gradient)
            means[(int64_t)f * 2 * C + which * C + c0 + c].v = mb;
            maed_agent_store(ex + (int64_t)f * 2 * C + which * C + c0 + c, bf2f(mb));
        }
        if (phase == 1) return;
        MAED_WAIT_VMCNT0();                                          // the published means have left this wave
        __syncthreads();
        if (S > 1) {
            if (tid == 0 && !maed_frame_arrive_and_wait(sync + (int64_t)f * 16, target)) { lfail = 1; maed_report_fault(fault); }
            __syncthreads();
        }
    }
    for (int i = tid; i < 2 * C; i += ST1_NT) lvec[i] = maed_agent_load(ex + (int64_t)f * 2 * C + i);
    __syncthreads();
    const float poison = lfail ? __uint_as_float(0x7fc00000u) : 0.f;
    st1_gemv256(w_ts, 2 * C, lvec, [=](int row) { return 2 * c0 + row; },
                [&](int row, float acc) { const float v = acc + b_ts[2 * c0 + row] + poison; logits[(int64_t)f * 2 * C + 2 * c0 + row] = v; llog[row] = v; });
    __syncthreads();
    float a0[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a0[j] = 1.f / (1.f + __expf(llog[2 * (cb * 8 + j) + 1] - llog[2 * (cb * 8 + j)]));     // softmax over the pair, weight of x_s
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int r = rsub + 32 * k;
        if (r >= P) continue;
        float a[8], b[8], o[8];
        st1_unpack(xs[k], a); st1_unpack(xt[k], b);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = b[j] * (1.f - a0[j]) + a[j] * a0[j];
        st8(mix + base + (int64_t)r * C, o);
    }
}

template <int CH>
// (P > 224: three tensors x 9 chunks do not fit 128 VGPRs -- one workgroup per CU there)
__global__ __launch_bounds__(ST1_NT, (CH > 7 ? 2 : 4)) void st_fused_bwd_kernel(const bf16* __restrict__ dmix, const bf16* __restrict__ x_s, const bf16* __restrict__ x_t,
                                                                 const float* __restrict__ logits, const bf16* __restrict__ wt_ts, bf16* __restrict__ dlogits,
                                                                 bf16* __restrict__ dx_s, bf16* __restrict__ dx_t, uint32_t* sync, float* ex, int P, int C,
                                                                 int S, uint32_t target, uint32_t* fault, int phase) {
    __shared__ float lpart[32 * 16 * 16];      // [row lane][chunk][sum dmix x_s: 8 | sum dmix x_t: 8]
    __shared__ float lvec[2048];               // the frame's 2C dlogits
    __shared__ float ldm[256];                 // d(means) of this slice: x_s half, x_t half
    __shared__ float la0[128];
    __shared__ int lfail;
    const int f = blockIdx.y, c0 = blockIdx.x * 128, tid = threadIdx.x, cb = tid & 15, rsub = tid >> 4;
    const int64_t base = (int64_t)f * P * C + c0 + cb * 8;
    uint4 dm[CH];
    if (tid == 0) lfail = 0;
    if (tid < 128) {
        const float l0 = logits[(int64_t)f * 2 * C + 2 * (c0 + tid)], l1 = logits[(int64_t)f * 2 * C + 2 * (c0 + tid) + 1];
        la0[tid] = 1.f / (1.f + __expf(l1 - l0));
    }
    {
        uint4 xs[CH], xt[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int r = rsub + 32 * k;
            if (r < P) {
                dm[k] = *reinterpret_cast<const uint4*>(dmix + base + (int64_t)r * C);
                if (phase != 2) { xs[k] = *reinterpret_cast<const uint4*>(x_s + base + (int64_t)r * C); xt[k] = *reinterpret_cast<const uint4*>(x_t + base + (int64_t)r * C); }
            } else { dm[k] = make_uint4(0u, 0u, 0u, 0u); xs[k] = dm[k]; xt[k] = dm[k]; }
        }
        if (phase != 2) {
            float s0[8], s1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                if (rsub + 32 * k >= P) continue;
                float d[8], a[8], b[8];
                st1_unpack(dm[k], d); st1_unpack(xs[k], a); st1_unpack(xt[k], b);
#pragma unroll
                for (int j = 0; j < 8; ++j) { s0[j] = fmaf(d[j], a[j], s0[j]); s1[j] = fmaf(d[j], b[j], s1[j]); }
            }
            float* mine = lpart + (rsub * 16 + cb) * 16;
#pragma unroll
            for (int j = 0; j < 8; ++j) { mine[j] = s0[j]; mine[8 + j] = s1[j]; }
        }
    }
    __syncthreads();
    if (phase != 2) {
        if (tid < 128) {
            const int c = tid;
            float d0 = 0.f, d1 = 0.f;
            for (int r = 0; r < 32; ++r) { d0 += lpart[(r * 16 + (c >> 3)) * 16 + (c & 7)]; d1 += lpart[(r * 16 + (c >> 3)) * 16 + 8 + (c & 7)]; }
            const float a0 = la0[c], a1 = 1.f - a0, dot = a0 * d0 + a1 * d1;
            const unsigned short g0 = f2bf(a0 * (d0 - dot)), g1 = f2bf(a1 * (d1 - dot));           // dlogits as the ts_attn GEMMs read them (bf16)
            const int64_t o = (int64_t)f * 2 * C + 2 * (c0 + c);
            dlogits[o].v = g0; dlogits[o + 1].v = g1;
            maed_agent_store(ex + o, bf2f(g0)); maed_agent_store(ex + o + 1, bf2f(g1));
        }
        if (phase == 1) return;
        MAED_WAIT_VMCNT0();
        __syncthreads();
        if (S > 1) {
            if (tid == 0 && !maed_frame_arrive_and_wait(sync + (int64_t)f * 16, target)) { lfail = 1; maed_report_fault(fault); }
            __syncthreads();
        }
    }
    for (int i = tid; i < 2 * C; i += ST1_NT) lvec[i] = maed_agent_load(ex + (int64_t)f * 2 * C + i);
    __syncthreads();
    const float poison = lfail ? __uint_as_float(0x7fc00000u) : 0.f;
    // d(means)[k] = sum_o dlogits[o] W_ts[o][k] = row k of the transposed weight image; rows 0..127: the x_s means of this slice, 128..255: the x_t means
    st1_gemv256(wt_ts, 2 * C, lvec, [=](int row) { return row < 128 ? c0 + row : C + c0 + (row - 128); },
                [&](int row, float acc) { ldm[row] = bf2f(f2bf(acc)) + poison; });                  // (bf16 like the stored d(means) of the separate path)
    __syncthreads();
    float a0[8], ms[8], mt[8];
    const float invP = 1.f / (float)P;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a0[j] = la0[cb * 8 + j]; ms[j] = ldm[cb * 8 + j] * invP; mt[j] = ldm[128 + cb * 8 + j] * invP; }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int r = rsub + 32 * k;
        if (r >= P) continue;
        float d[8], os[8], ot[8];
        st1_unpack(dm[k], d);
#pragma unroll
        for (int j = 0; j < 8; ++j) { os[j] = d[j] * a0[j] + ms[j]; ot[j] = d[j] * (1.f - a0[j]) + mt[j]; }
        st8(dx_s + base + (int64_t)r * C, os); st8(dx_t + base + (int64_t)r * C, ot);
    }
}

// (after a frame-barrier timeout in this process -- maed_device_faults() -- the answer is no: callers fall back to the four-launch sequence)
extern "C" int maed_st_fused_supported(int P, int C, int dtype) {
    return dtype == MAED_BF16 && C % 128 == 0 && 2 * C <= 2048 && P > 0 && P <= 288 && !maed_fault_seen("st_fused");
}

#ifdef MAED_HOSTSIM
#define ST1_LAUNCH(K_, ...) do { hipLaunchKernelGGL(K_, grid, dim3(ST1_NT), 0, s, __VA_ARGS__, 1); hipLaunchKernelGGL(K_, grid, dim3(ST1_NT), 0, s, __VA_ARGS__, 2); } while (0)
#else
#define ST1_LAUNCH(K_, ...) hipLaunchKernelGGL(K_, grid, dim3(ST1_NT), 0, s, __VA_ARGS__, 0)
#endif

// arrive_base: value of the frames' arrival counters at launch -- 0 after a clear; the block driver's backward continues from the forward's C / 128 instead of
// clearing again (one counter, monotonic over the two launches of a block); clear_sync: memset the counters first (stand-alone calls)
int maed_st_fused_fwd_ws(const void* x_s, const void* x_t, const void* w_ts, const float* b_ts, void* means, float* logits, void* mix,
                         uint32_t* sync, float* ex, int F, int P, int C, int dtype, bool clear_sync, uint32_t arrive_base, void* stream);
int maed_st_fused_bwd_ws(const void* dmix, const void* x_s, const void* x_t, const float* logits, const void* wt_ts, void* dlogits, void* dx_s,
                         void* dx_t, uint32_t* sync, float* ex, int F, int P, int C, int dtype, bool clear_sync, uint32_t arrive_base, void* stream);
extern "C" int maed_st_fused_fwd(const void* x_s, const void* x_t, const void* w_ts, const float* b_ts, void* means, float* logits, void* mix,
                                 uint32_t* sync, float* ex, int F, int P, int C, int dtype, void* stream) {
    return maed_st_fused_fwd_ws(x_s, x_t, w_ts, b_ts, means, logits, mix, sync, ex, F, P, C, dtype, true, 0u, stream);
}
extern "C" int maed_st_fused_bwd(const void* dmix, const void* x_s, const void* x_t, const float* logits, const void* wt_ts, void* dlogits, void* dx_s,
                                 void* dx_t, uint32_t* sync, float* ex, int F, int P, int C, int dtype, void* stream) {
    return maed_st_fused_bwd_ws(dmix, x_s, x_t, logits, wt_ts, dlogits, dx_s, dx_t, sync, ex, F, P, C, dtype, true, 0u, stream);
}
int maed_st_fused_fwd_ws(const void* x_s, const void* x_t, const void* w_ts, const float* b_ts, void* means, float* logits, void* mix,
                         uint32_t* sync, float* ex, int F, int P, int C, int dtype, bool clear_sync, uint32_t arrive_base, void* stream) {
    MAED_CHECK_ARG(x_s && x_t && w_ts && b_ts && means && logits && mix && sync && ex, MAED_ERR_ARG, "st_fused_fwd: null pointer");
    MAED_CHECK_ARG(maed_st_fused_supported(P, C, dtype), MAED_ERR_UNSUPPORTED, "st_fused_fwd: bf16, C %% 128 == 0, C <= 1024, P <= 288 (P=%d C=%d dtype=%d)", P, C, dtype);
    MAED_CHECK_ARG(is_aligned(x_s, 16) && is_aligned(x_t, 16) && is_aligned(w_ts, 16) && is_aligned(mix, 16), MAED_ERR_ALIGN, "st_fused_fwd: 16-B alignment");
    if (F <= 0) return MAED_OK;
    hipStream_t s = (hipStream_t)stream;
    if (clear_sync) MAED_HIP(hipMemsetAsync(sync, 0, (size_t)F * 16 * sizeof(uint32_t), s), "st_fused_fwd: memset");
    const int S = C / 128;
    const uint32_t target = arrive_base + (uint32_t)S;
    const dim3 grid(S, F);
    if (P <= 224) ST1_LAUNCH(st_fused_fwd_kernel<7>, (const bf16*)x_s, (const bf16*)x_t, (const bf16*)w_ts, b_ts, (bf16*)means, logits, (bf16*)mix, sync, ex, P, C, S, target, maed_fault_word());
    else ST1_LAUNCH(st_fused_fwd_kernel<9>, (const bf16*)x_s, (const bf16*)x_t, (const bf16*)w_ts, b_ts, (bf16*)means, logits, (bf16*)mix, sync, ex, P, C, S, target, maed_fault_word());
    MAED_CHECK_LAUNCH("st_fused_fwd");
    return MAED_OK;
}

int maed_st_fused_bwd_ws(const void* dmix, const void* x_s, const void* x_t, const float* logits, const void* wt_ts, void* dlogits, void* dx_s,
                         void* dx_t, uint32_t* sync, float* ex, int F, int P, int C, int dtype, bool clear_sync, uint32_t arrive_base, void* stream) {
    MAED_CHECK_ARG(dmix && x_s && x_t && logits && wt_ts && dlogits && dx_s && dx_t && sync && ex, MAED_ERR_ARG, "st_fused_bwd: null pointer");
    MAED_CHECK_ARG(maed_st_fused_supported(P, C, dtype), MAED_ERR_UNSUPPORTED, "st_fused_bwd: bf16, C %% 128 == 0, C <= 1024, P <= 288 (P=%d C=%d dtype=%d)", P, C, dtype);
    MAED_CHECK_ARG(is_aligned(dmix, 16) && is_aligned(x_s, 16) && is_aligned(x_t, 16) && is_aligned(wt_ts, 16) && is_aligned(dx_s, 16) && is_aligned(dx_t, 16),
                   MAED_ERR_ALIGN, "st_fused_bwd: 16-B alignment");
    if (F <= 0) return MAED_OK;
    hipStream_t s = (hipStream_t)stream;
    if (clear_sync) MAED_HIP(hipMemsetAsync(sync, 0, (size_t)F * 16 * sizeof(uint32_t), s), "st_fused_bwd: memset");
    const int S = C / 128;
    const uint32_t target = arrive_base + (uint32_t)S;
    const dim3 grid(S, F);
    if (P <= 224) ST1_LAUNCH(st_fused_bwd_kernel<7>, (const bf16*)dmix, (const bf16*)x_s, (const bf16*)x_t, logits, (const bf16*)wt_ts, (bf16*)dlogits, (bf16*)dx_s,
                             (bf16*)dx_t, sync, ex, P, C, S, target, maed_fault_word());
    else ST1_LAUNCH(st_fused_bwd_kernel<9>, (const bf16*)dmix, (const bf16*)x_s, (const bf16*)x_t, logits, (const bf16*)wt_ts, (bf16*)dlogits, (bf16*)dx_s,
                    (bf16*)dx_t, sync, ex, P, C, S, target, maed_fault_word());
    MAED_CHECK_LAUNCH("st_fused_bwd");
    return MAED_OK;
}
#undef ST1_LAUNCH

// the token-axis reductions are split CM_SPLIT ways; partial sums meet in a caller-owned fp32 scratch `ws`
extern "C" int maed_st_colmean(const void* x_s, const void* x_t, void* means, float* ws /* F*2C fp32 */, int F, int P, int C,
                                  int dtype, void* stream) {
    MAED_CHECK_ARG(x_s && x_t && means && ws, MAED_ERR_ARG, "st_colmean: null pointer");
    MAED_CHECK_ARG(C % 4 == 0 && P > 0, MAED_ERR_SHAPE, "st_colmean: C %% 4");
    if (F == 0) return MAED_OK;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)F * 2 * C;
    MAED_HIP(hipMemsetAsync(ws, 0, n * sizeof(float), s), "memset");
    dim3 grid((C / 4 + 63) / 64, F, 2 * CM_SPLIT);
    MAED_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL((st_colmean_kernel<T>), grid, dim3(64), 0, s, (const T*)x_s, (const T*)x_t, ws, P, C);
        hipLaunchKernelGGL((st_colmean_finish<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ws, (T*)means, n, 1.f / (float)P);
    });
    MAED_CHECK_LAUNCH("st_colmean");
    return MAED_OK;
}

extern "C" int maed_st_mix_fwd(const void* x_s, const void* x_t, const float* logits, void* mix, int F, int P, int C, int dtype,
                               void* stream) {
    MAED_CHECK_ARG(x_s && x_t && logits && mix, MAED_ERR_ARG, "st_mix_fwd: null pointer");
    MAED_CHECK_ARG(C % 4 == 0, MAED_ERR_SHAPE, "st_mix_fwd: C %% 4");
    const int64_t n4 = (int64_t)F * P * C / 4;
    if (n4 == 0) return MAED_OK;
    MAED_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((st_mix_fwd_kernel<T>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                                                      (const T*)x_s, (const T*)x_t, logits, (T*)mix, n4, P, C));
    MAED_CHECK_LAUNCH("st_mix_fwd");
    return MAED_OK;
}

extern "C" int maed_st_mix_bwd_reduce(const void* dmix, const void* x_s, const void* x_t, const float* logits, void* dlogits,
                                         float* ws /* F*2C fp32 */, int F, int P, int C, int dtype, void* stream) {
    MAED_CHECK_ARG(dmix && x_s && x_t && logits && dlogits && ws, MAED_ERR_ARG, "st_mix_bwd_reduce: null pointer");
    MAED_CHECK_ARG(C % 4 == 0, MAED_ERR_SHAPE, "st_mix_bwd_reduce: C %% 4");
    if (F == 0) return MAED_OK;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)F * 2 * C;
    MAED_HIP(hipMemsetAsync(ws, 0, n * sizeof(float), s), "memset");
    dim3 grid((C / 4 + 63) / 64, F, CM_SPLIT);
    MAED_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL((st_mix_bwd_reduce_kernel<T>), grid, dim3(64), 0, s, (const T*)dmix, (const T*)x_s, (const T*)x_t, ws, P, C);
        hipLaunchKernelGGL((st_mix_bwd_reduce_finish<T>), dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, s, ws, logits, (T*)dlogits, n / 2);
    });
    MAED_CHECK_LAUNCH("st_mix_bwd_reduce");
    return MAED_OK;
}

extern "C" int maed_st_mix_bwd_apply(const void* dmix, const float* logits, const void* dmeans, void* dx_s, void* dx_t, int F, int P,
                                     int C, int dtype, void* stream) {
    MAED_CHECK_ARG(dmix && logits && dmeans && dx_s && dx_t, MAED_ERR_ARG, "st_mix_bwd_apply: null pointer");
    MAED_CHECK_ARG(C % 4 == 0, MAED_ERR_SHAPE, "st_mix_bwd_apply: C %% 4");
    const int64_t n4 = (int64_t)F * P * C / 4;
    if (n4 == 0) return MAED_OK;
    MAED_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((st_mix_bwd_apply_kernel<T>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                                                      (const T*)dmix, logits, (const T*)dmeans, (T*)dx_s, (T*)dx_t, n4, P, C));
    MAED_CHECK_LAUNCH("st_mix_bwd_apply");
    return MAED_OK;
}

// ---------------------------------------------------------------------------------------------------
// K8 embeddings (vision_transformer.py:392-399)
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const T* __restrict__ patch, const float* __restrict__ cls, const float* __restrict__ pos,
                                                        const float* __restrict__ temp, float* __restrict__ tok, int64_t n4, int P, int C, int Tn) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int64_t e = i * 4;
    const int c = (int)(e % C); const int64_t r = e / C;
    const int p = (int)(r % P); const int64_t f = r / P;
    float a[4], b[4], t[4], o[4];
    if (p == 0) ld4(cls + c, a); else ld4(patch + (f * (P - 1) + (p - 1)) * C + c, a);
    ld4(pos + (int64_t)p * C + c, b);
    ld4(temp + (f % Tn) * C + c, t);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (a[j] + b[j]) + t[j];  // same association as the reference: (x + pos) + temp
    st4(tok + e, o);
}

// thread per (p, 4 channels): dpatch rows copied/cast, dpos[p] += sum_f dtok[f][p]
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_pos_kernel(const float* __restrict__ dtok, T* __restrict__ dpatch, float* __restrict__ dpos,
                                                            int F, int P, int C) {
    // (round 4: the frames are split EMB_SPLIT ways over blockIdx.y and meet in dpos with atomics -- one thread per (p, 4 channels) walking all 128 frames was 99
    //  workgroups with one dependent load each per step: 59 us for 77 MB)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n4 = (int64_t)P * C / 4;
    if (i >= n4) return;
    const int64_t e = i * 4;
    const int c = (int)(e % C); const int p = (int)(e / C);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int f = blockIdx.y; f < F; f += gridDim.y) {
        float v[4];
        ld4(dtok + ((int64_t)f * P + p) * C + c, v);
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        if (p > 0 && dpatch) st4(dpatch + ((int64_t)f * (P - 1) + (p - 1)) * C + c, v);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(dpos + e + j, s[j]);
}
// thread per (f, 4 channels): dtemp[f % T] += sum_p dtok[f][p]   (the temporal embedding is shared by the frames at the same position of every clip)
__global__ __launch_bounds__(256) void embed_bwd_frame_kernel(const float* __restrict__ dtok, float* __restrict__ dtemp, int F, int P, int C, int T) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n4 = (int64_t)F * C / 4;
    if (i >= n4) return;
    const int64_t e = i * 4;
    const int c = (int)(e % C); const int64_t f = e / C;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int p = blockIdx.y; p < P; p += gridDim.y) {          // (tokens split over blockIdx.y, see above)
        float v[4];
        ld4(dtok + (f * P + p) * C + c, v);
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    float* dst = dtemp + (f % T) * (int64_t)C + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(dst + j, s[j]);
}

extern "C" int maed_embed_add_fwd(const void* patch, int dtype, const float* cls, const float* pos, const float* temp, float* tokens,
                                  int F, int P, int C, int T, void* stream) {
    MAED_CHECK_ARG(patch && cls && pos && temp && tokens, MAED_ERR_ARG, "embed_add_fwd: null pointer");
    MAED_CHECK_ARG(C % 4 == 0 && T > 0 && P > 1, MAED_ERR_SHAPE, "embed_add_fwd: bad extents");
    const int64_t n4 = (int64_t)F * P * C / 4;
    if (n4 == 0) return MAED_OK;
    MAED_DISPATCH_DTYPE(dtype, TT, hipLaunchKernelGGL((embed_fwd_kernel<TT>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                                                       (const TT*)patch, cls, pos, temp, tokens, n4, P, C, T));
    MAED_CHECK_LAUNCH("embed_add_fwd");
    return MAED_OK;
}

extern "C" int maed_embed_add_bwd(const float* dtokens, void* dpatch, int dtype, float* dpos, float* dtemp, int F, int P, int C, int T,
                                  void* stream) {
    MAED_CHECK_ARG(dtokens && dpos && dtemp, MAED_ERR_ARG, "embed_add_bwd: null pointer");
    MAED_CHECK_ARG(C % 4 == 0 && P > 1 && T > 0, MAED_ERR_SHAPE, "embed_add_bwd: bad extents");
    if (F == 0) return MAED_OK;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n4 = (int64_t)P * C / 4, m4 = (int64_t)F * C / 4;
#define EMB_SPLIT 8
    MAED_DISPATCH_DTYPE(dtype, TT, hipLaunchKernelGGL((embed_bwd_pos_kernel<TT>), dim3((unsigned)((n4 + 255) / 256), F < EMB_SPLIT ? F : EMB_SPLIT), dim3(256), 0, s,
                                                       dtokens, (TT*)dpatch, dpos, F, P, C));
    hipLaunchKernelGGL(embed_bwd_frame_kernel, dim3((unsigned)((m4 + 255) / 256), P < EMB_SPLIT ? P : EMB_SPLIT), dim3(256), 0, s, dtokens, dtemp, F, P, C, T);
#undef EMB_SPLIT
    MAED_CHECK_LAUNCH("embed_add_bwd");
    return MAED_OK;
}

// ---------------------------------------------------------------------------------------------------
// transpose + cast (+ column sums): out_t[n][m] = in[m][n]
// ---------------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const TI* __restrict__ in, int64_t ldi, int64_t M, int64_t N, TO* __restrict__ out_t,
                                                             int64_t ldt, TO* __restrict__ out_c, int64_t ldc, float* __restrict__ colsum) {
    __shared__ float tile[64][65];
    const int64_t m0 = (int64_t)blockIdx.y * 64, n0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 4 rows of 64 per pass
    for (int r = ty; r < 64; r += 4) {
        const int64_t m = m0 + r, n = n0 + tx;
        float v = 0.f;
        if (m < M && n < N) {
            v = ldf(in + m * ldi + n);
            if (out_c) stf(out_c + m * ldc + n, v);
        }
        tile[r][tx] = v;
    }
    if (colsum) {  // combine the 4 partial sums of each column through LDS after the tile is complete
        __syncthreads();
        if (ty == 0) {
            float s = 0.f;
            for (int r = 0; r < 64; ++r) s += tile[r][tx];
            if (n0 + tx < N) atomicAdd(colsum + n0 + tx, s);
        }
    } else {
        __syncthreads();
    }
    if (out_t) {
        for (int r = ty; r < 64; r += 4) {  // r indexes n, tx indexes m
            const int64_t n = n0 + r, m = m0 + tx;
            if (n < N && m < ldt) stf(out_t + n * ldt + m, tile[tx][r]);  // rows m >= M were zero-filled
        }
    }
}

extern "C" int maed_transpose_cast(const void* in, int in_dtype, int64_t ldi, int64_t M, int64_t N, void* out_t, int64_t ldt,
                                   void* out_c, int64_t ldc, float* colsum, int dtype, void* stream) {
    MAED_CHECK_ARG(in, MAED_ERR_ARG, "transpose_cast: null input");
    MAED_CHECK_ARG(M > 0 && N > 0 && ldi >= N && (!out_t || ldt >= M) && (!out_c || ldc >= N), MAED_ERR_SHAPE, "transpose_cast: bad extents");
    const int64_t rows = out_t ? ldt : M;  // cover the zero-filled padding columns of out_t
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((rows + 63) / 64));
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == MAED_F32 && dtype == MAED_F32)
        hipLaunchKernelGGL((transpose_cast_kernel<float, float>), grid, dim3(256), 0, s, (const float*)in, ldi, M, N, (float*)out_t, ldt, (float*)out_c, ldc, colsum);
    else if (in_dtype == MAED_F32 && dtype == MAED_BF16)
        hipLaunchKernelGGL((transpose_cast_kernel<float, bf16>), grid, dim3(256), 0, s, (const float*)in, ldi, M, N, (bf16*)out_t, ldt, (bf16*)out_c, ldc, colsum);
    else if (in_dtype == MAED_BF16 && dtype == MAED_BF16)
        hipLaunchKernelGGL((transpose_cast_kernel<bf16, bf16>), grid, dim3(256), 0, s, (const bf16*)in, ldi, M, N, (bf16*)out_t, ldt, (bf16*)out_c, ldc, colsum);
    else { maed_set_error("transpose_cast: unsupported dtype pair %d -> %d", in_dtype, dtype); return MAED_ERR_UNSUPPORTED; }
    MAED_CHECK_LAUNCH("transpose_cast");
    return MAED_OK;
}

// ---------------------------------------------------------------------------------------------------
// nn.Dropout(p) in training (ktd.py:54,56: the KTD's two Dropout(0.5) layers are the only active dropouts on the path) and
// tanh' (pre_logits, vision_transformer.py:350-353) -- the two element-wise pieces of the training tail that used to be ATen ops.
// The keep mask is a counter-based hash of (seed, element index): nothing is stored, the backward recomputes it from the seed.
// (The reference draws its mask from torch's Philox stream on whatever device it runs on; no two devices share that stream, so
// the contract is the distribution: keep probability 1-p, survivors scaled by 1/(1-p).)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t idx, uint32_t thresh24) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;                 // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 40) >= thresh24;                                // top 24 bits uniform in [0, 2^24)
}

// y = keep ? x * scale : 0     (the same kernel is its own backward: dx = keep ? dy * scale : 0)
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, uint64_t seed, uint32_t thresh24,
                                                      float scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = dropout_keep(seed, (uint64_t)i, thresh24) ? x[i] * scale : 0.f;
}

extern "C" int maed_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream) {
    MAED_CHECK_ARG(x && y, MAED_ERR_ARG, "dropout: null pointer");
    MAED_CHECK_ARG(n >= 0 && p >= 0.f && p < 1.f, MAED_ERR_ARG, "dropout: need 0 <= p < 1 (p=%f)", (double)p);
    if (n == 0) return MAED_OK;
    const uint32_t thresh = (uint32_t)((double)p * 16777216.0);
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, seed, thresh, 1.0f / (1.0f - p));
    MAED_CHECK_LAUNCH("dropout");
    return MAED_OK;
}

// the same with the seed in DEVICE memory (maed_train_state, round 6): a captured hipGraph replays with fresh masks -- the host refreshes the state before every replay,
// the launch arguments never change.  call_id separates the Dropout layers of one step (and ties a layer's backward to its forward).
__global__ __launch_bounds__(256) void dropout_dev_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, const maed_train_state* __restrict__ st,
                                                          uint64_t call_id, uint32_t thresh24, float scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t seed = st->seed + call_id * 0x9E3779B97F4A7C15ull;
    if (i < n) y[i] = dropout_keep(seed, (uint64_t)i, thresh24) ? x[i] * scale : 0.f;
}

extern "C" int maed_dropout_dev(const float* x, float* y, int64_t n, float p, const maed_train_state* state, uint64_t call_id, void* stream) {
    MAED_CHECK_ARG(x && y && state, MAED_ERR_ARG, "dropout_dev: null pointer");
    MAED_CHECK_ARG(n >= 0 && p >= 0.f && p < 1.f, MAED_ERR_ARG, "dropout_dev: need 0 <= p < 1 (p=%f)", (double)p);
    MAED_CHECK_ARG(is_aligned(state, 8), MAED_ERR_ALIGN, "dropout_dev: state must be 8-B aligned");
    if (n == 0) return MAED_OK;
    const uint32_t thresh = (uint32_t)((double)p * 16777216.0);
    hipLaunchKernelGGL(dropout_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, state, call_id, thresh, 1.0f / (1.0f - p));
    MAED_CHECK_LAUNCH("dropout_dev");
    return MAED_OK;
}

// dx = dy * (1 - y^2)   (y = tanh(.) as the GEMM's TANH epilogue stored it; T = its storage type)
template <typename T>
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const float t = ldf(y + i); stf(dx + i, dy[i] * (1.0f - t * t)); }
}

extern "C" int maed_tanh_bwd(const float* dy, const void* y, void* dx, int64_t n, int dtype, void* stream) {
    MAED_CHECK_ARG(dy && y && dx, MAED_ERR_ARG, "tanh_bwd: null pointer");
    if (n <= 0) return MAED_OK;
    MAED_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((tanh_bwd_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, (const T*)y, (T*)dx, n));
    MAED_CHECK_LAUNCH("tanh_bwd");
    return MAED_OK;
}

// ---------------------------------------------------------------------------------------------------
// batched weight refresh: for every fp32 master weight W_i (rows_i x cols_i) of the table write the compute-dtype copy and the
// transposed compute-dtype copy in ONE launch (after an optimizer step the STE needs both images of 32 matrices: that was 32
// launches of the kernel above)
// ---------------------------------------------------------------------------------------------------
struct WtEntry { const float* src; void* dst_c; void* dst_t; int32_t rows, cols, tile0, tiles_n; };   // tile0 = first 64x64 tile of this matrix
#define WT_MAX_ENTRIES 256

template <typename TO>
__global__ __launch_bounds__(256) void weight_refresh_kernel(const WtEntry* __restrict__ tab, int n_entries) {
    __shared__ float tile[64][65];
    int e = 0;
    for (int i = 1; i < n_entries; ++i) e = ((int)blockIdx.x >= tab[i].tile0) ? i : e;     // block-uniform, table is tiny
    const WtEntry d = tab[e];
    const int t = blockIdx.x - d.tile0;
    const int m0 = (t / d.tiles_n) * 64, n0 = (t % d.tiles_n) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    TO* out_c = (TO*)d.dst_c;
    TO* out_t = (TO*)d.dst_t;
    if (m0 + 64 <= d.rows && n0 + 64 <= d.cols && (d.rows & 3) == 0 && (d.cols & 3) == 0) {
        // full tile of a matrix with 4-element-aligned extents (every nn.Linear of the model): 16-byte loads, 4-element stores on both images (the scalar path below
        // ran at 1.9 TB/s: 82 us per step)
        const int cg = (threadIdx.x & 15) * 4, r0 = threadIdx.x >> 4;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int r = r0 + 16 * ps;
            float v[4];
            ld4(d.src + (int64_t)(m0 + r) * d.cols + n0 + cg, v);
            if (out_c) st4(out_c + (int64_t)(m0 + r) * d.cols + n0 + cg, v);
            tile[r][cg] = v[0]; tile[r][cg + 1] = v[1]; tile[r][cg + 2] = v[2]; tile[r][cg + 3] = v[3];
        }
        __syncthreads();
        if (out_t) {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int n = r0 + 16 * ps;                                   // row of the transposed image
                const float v[4] = {tile[cg][n], tile[cg + 1][n], tile[cg + 2][n], tile[cg + 3][n]};
                st4(out_t + (int64_t)(n0 + n) * d.rows + m0 + cg, v);
            }
        }
        return;
    }
    for (int r = ty; r < 64; r += 4) {
        const int m = m0 + r, n = n0 + tx;
        float v = 0.f;
        if (m < d.rows && n < d.cols) {
            v = d.src[(int64_t)m * d.cols + n];
            if (out_c) stf(out_c + (int64_t)m * d.cols + n, v);
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    if (out_t)
        for (int r = ty; r < 64; r += 4) {
            const int n = n0 + r, m = m0 + tx;
            if (n < d.cols && m < d.rows) stf(out_t + (int64_t)n * d.rows + m, tile[tx][r]);
        }
}

extern "C" int maed_weight_refresh(const void* table, int n_entries, int n_tiles, int dtype, void* stream) {
    MAED_CHECK_ARG(table, MAED_ERR_ARG, "weight_refresh: null table");
    MAED_CHECK_ARG(n_entries > 0 && n_entries <= WT_MAX_ENTRIES, MAED_ERR_SHAPE, "weight_refresh: n_entries=%d out of range", n_entries);
    if (n_tiles <= 0) return MAED_OK;
    MAED_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((weight_refresh_kernel<T>), dim3(n_tiles), dim3(256), 0, (hipStream_t)stream,
                                                      (const WtEntry*)table, n_entries));
    MAED_CHECK_LAUNCH("weight_refresh");
    return MAED_OK;
}

// ---------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam semantics: L2 weight decay folded into the gradient)
// ---------------------------------------------------------------------------------------------------
// st != NULL (maed_adam_step_dev): learning rate and bias corrections come from DEVICE memory, written by the host before the launch runs -- the launch arguments of a
// captured step never change
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   bf16* __restrict__ shadow, int64_t n, float lr, float b1, float b2, float eps, float wd,
                                                   float bc1, float bc2, float gscale, const maed_train_state* __restrict__ st) {
    const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (st) { lr = st->lr; bc1 = st->bias_corr1; bc2 = st->bias_corr2; }
    const float step = lr / bc1, rbc2 = rsqrtf(bc2);
    if (i4 + 4 <= n) {
        float pp[4], gg[4], mm[4], vv[4];
        ld4(p + i4, pp); ld4(g + i4, gg); ld4(m + i4, mm); ld4(v + i4, vv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gr = fmaf(wd, pp[j], gg[j] * gscale);
            mm[j] = b1 * mm[j] + (1.f - b1) * gr;
            vv[j] = b2 * vv[j] + (1.f - b2) * gr * gr;
            pp[j] -= step * mm[j] / (sqrtf(vv[j]) * rbc2 + eps);
        }
        st4(p + i4, pp); st4(m + i4, mm); st4(v + i4, vv);
        if (shadow) st4(shadow + i4, pp);
    } else {
        for (int64_t i = i4; i < n; ++i) {
            const float gr = fmaf(wd, p[i], g[i] * gscale);
            m[i] = b1 * m[i] + (1.f - b1) * gr;
            v[i] = b2 * v[i] + (1.f - b2) * gr * gr;
            p[i] -= step * m[i] / (sqrtf(v[i]) * rbc2 + eps);
            if (shadow) shadow[i].v = f2bf(p[i]);
        }
    }
}

extern "C" int maed_adam_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2, float gscale,
                              void* stream) {
    MAED_CHECK_ARG(p && g && m && v, MAED_ERR_ARG, "adam_step: null pointer");
    MAED_CHECK_ARG(is_aligned(p, 16) && is_aligned(g, 16) && is_aligned(m, 16) && is_aligned(v, 16) && (!shadow_bf16 || is_aligned(shadow_bf16, 8)),
                   MAED_ERR_ALIGN, "adam_step: arenas must be 16-B aligned");
    if (n == 0) return MAED_OK;
    const int64_t nthr = (n + 3) / 4;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16*)shadow_bf16, n, lr,
                       beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, gscale, (const maed_train_state*)nullptr);
    MAED_CHECK_LAUNCH("adam_step");
    return MAED_OK;
}

extern "C" int maed_adam_step_dev(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, const maed_train_state* state, float beta1,
                                  float beta2, float eps, float weight_decay, float gscale, void* stream) {
    MAED_CHECK_ARG(p && g && m && v && state, MAED_ERR_ARG, "adam_step_dev: null pointer");
    MAED_CHECK_ARG(is_aligned(p, 16) && is_aligned(g, 16) && is_aligned(m, 16) && is_aligned(v, 16) && (!shadow_bf16 || is_aligned(shadow_bf16, 8)) && is_aligned(state, 8),
                   MAED_ERR_ALIGN, "adam_step_dev: arenas must be 16-B aligned, the state 8-B");
    if (n == 0) return MAED_OK;
    const int64_t nthr = (n + 3) / 4;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16*)shadow_bf16, n, 0.f,
                       beta1, beta2, eps, weight_decay, 1.f, 1.f, gscale, state);
    MAED_CHECK_LAUNCH("adam_step_dev");
    return MAED_OK;
}
