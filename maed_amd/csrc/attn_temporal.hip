// K4: temporal attention of the STE (vision_transformer.py:216-228): per (clip n, head h, token p)
//     attention across the T frames of the clip.  The reference makes three permuted copies of
//     q/k/v and one of the output per block; here every thread gathers its rows straight from the
//     (F,P,3C) qkv buffer (frame stride P*3C) and writes the (F,P,C) result in place.
// HBM-bound (arithmetic intensity ~T/2 flop/B): thread per (n,h,p,t) query row, the T key/value rows
// of the same (n,h,p) are shared by T neighbouring threads through L1.  fp32 math for both dtypes.
#include "common.cuh"
#include "attn_mfma.cuh"
#include "gemm_x3.h"
#include <stdlib.h>

#define D HEAD_DIM

template <typename T>
__device__ __forceinline__ void load_row(const T* p, float (&v)[D]) {
#pragma unroll
    for (int c = 0; c < D; c += 8) {
        float t[8];
        ld8(p + c, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[c + j] = t[j];
    }
}
template <typename T>
__device__ __forceinline__ float dot_row(const T* p, const float (&v)[D]) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D; c += 8) {
        float t[8];
        ld8(p + c, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) s = fmaf(v[c + j], t[j], s);
    }
    return s;
}
template <typename T>
__device__ __forceinline__ void axpy_row(const T* p, float a, float (&acc)[D]) {
#pragma unroll
    for (int c = 0; c < D; c += 8) {
        float t[8];
        ld8(p + c, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[c + j] = fmaf(a, t[j], acc[c + j]);
    }
}
template <typename T>
__device__ __forceinline__ void store_row(T* p, const float (&v)[D], float mul, int accumulate) {
#pragma unroll
    for (int c = 0; c < D; c += 8) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = v[c + j] * mul;
        if (accumulate) { float o[8]; ld8(p + c, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] += o[j]; }
        st8(p + c, t);
    }
}

// gid = ((n*H + h)*P + p)*T + t
template <typename T>
__global__ __launch_bounds__(256) void attn_tm_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ o, float* __restrict__ lse,
                                                          int64_t total, int P, int H, int Tn, float scale) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int t = (int)(gid % Tn); int64_t r = gid / Tn;
    const int p = (int)(r % P); r /= P;
    const int h = (int)(r % H); const int64_t n = r / H;
    const int C = H * D; const int64_t ld = 3 * (int64_t)C;
    const int64_t f = n * Tn + t;
    const T* qrow = qkv + (f * P + p) * ld + h * D;
    float qv[D], acc[D];
    load_row(qrow, qv);
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = 0.f;
    float m = -INFINITY, l = 0.f;
    for (int t2 = 0; t2 < Tn; ++t2) {
        const T* krow = qkv + ((n * Tn + t2) * P + p) * ld + C + h * D;
        const float s = dot_row(krow, qv) * scale;
        const float mn = fmaxf(m, s);
        const float a = __expf(m - mn), pr = __expf(s - mn);
        l = l * a + pr;
#pragma unroll
        for (int c = 0; c < D; ++c) acc[c] *= a;
        axpy_row(krow + C, pr, acc);
        m = mn;
    }
    store_row(o + (f * P + p) * C + h * D, acc, 1.f / l, 0);
    lse[(f * H + h) * P + p] = m + __logf(l);
}

template <typename T>
__global__ __launch_bounds__(256) void attn_tm_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ o, const T* __restrict__ d_o,
                                                          const float* __restrict__ lse, T* __restrict__ dqkv, int accumulate,
                                                          int64_t total, int P, int H, int Tn, float scale) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int t = (int)(gid % Tn); int64_t r = gid / Tn;
    const int p = (int)(r % P); r /= P;
    const int h = (int)(r % H); const int64_t n = r / H;
    const int C = H * D; const int64_t ld = 3 * (int64_t)C;
    const int64_t f = n * Tn + t;
    const int64_t row = f * P + p;
    float a[D], b[D], acc[D];
    // ---- as query t: dQ = sum_t2 ds[t][t2] K[t2] ----
    load_row(qkv + row * ld + h * D, a);          // q
    load_row(d_o + row * C + h * D, b);           // dO
    const float Dq = dot_row(o + row * C + h * D, b);
    const float L = lse[(f * H + h) * P + p];
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = 0.f;
    for (int t2 = 0; t2 < Tn; ++t2) {
        const T* krow = qkv + ((n * Tn + t2) * P + p) * ld + C + h * D;
        const float pr = __expf(dot_row(krow, a) * scale - L);
        const float ds = pr * (dot_row(krow + C, b) - Dq) * scale;
        axpy_row(krow, ds, acc);
    }
    store_row(dqkv + row * ld + h * D, acc, 1.f, accumulate);
    // ---- as key t: dV = sum_t1 p[t1][t] dO[t1] ----
    load_row(qkv + row * ld + C + h * D, a);      // k
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = 0.f;
    for (int t1 = 0; t1 < Tn; ++t1) {
        const int64_t row1 = (n * Tn + t1) * P + p;
        const float pr = __expf(dot_row(qkv + row1 * ld + h * D, a) * scale - lse[((n * Tn + t1) * H + h) * P + p]);
        axpy_row(d_o + row1 * C + h * D, pr, acc);
    }
    store_row(dqkv + row * ld + 2 * C + h * D, acc, 1.f, accumulate);
    // ---- as key t: dK = sum_t1 ds[t1][t] Q[t1] ----
    const T* vrow = qkv + row * ld + 2 * C + h * D;  // v[t] stays in L1; registers hold k, dO[t1], acc
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = 0.f;
    for (int t1 = 0; t1 < Tn; ++t1) {
        const int64_t row1 = (n * Tn + t1) * P + p;
        const T* q1 = qkv + row1 * ld + h * D;
        const T* do1 = d_o + row1 * C + h * D;
        const float pr = __expf(dot_row(q1, a) * scale - lse[((n * Tn + t1) * H + h) * P + p]);
        float d1[D];
        load_row(do1, d1);
        const float dp = dot_row(vrow, d1);  // dO[t1] . v[t]
        const float D1 = dot_row(o + row1 * C + h * D, d1);
        const float ds = pr * (dp - D1) * scale;
        axpy_row(q1, ds, acc);
    }
    store_row(dqkv + row * ld + C + h * D, acc, 1.f, accumulate);
}

// ==================================================================================================
// LDS-staged variants (used when T divides the workgroup): a workgroup owns GP token positions x T frames
// of one (clip, head); the K/V (and for the backward Q/dO) rows of those GP*T (frame, token) pairs are
// staged ONCE in LDS with fully coalesced 128-B line loads, then every thread (gp, t) walks the T rows of
// its group from LDS (broadcast reads; a 16-B skew per group keeps the groups of a wave on distinct banks).
// HBM traffic = the algorithmic minimum (qkv read once, o written once).
// ==================================================================================================
template <typename T> struct RowGeom { static constexpr int RS = D * (int)sizeof(T); };  // row bytes

template <typename T>
__device__ __forceinline__ T* lds_row(char* base, int r, int gp) { return reinterpret_cast<T*>(base + (size_t)r * RowGeom<T>::RS + gp * 16); }

// cooperative stage: rows r = gp*Tn + t  <-  src(frame n*Tn+t, token p0+gp) ; 16-B chunks, 8|16 per row
template <typename T>
__device__ __forceinline__ void stage_group_rows(char* dst, const T* src_base, int64_t ld, int64_t n, int Tn, int P, int p0, int GP, int nthr) {
    constexpr int CH = RowGeom<T>::RS / 16;         // chunks per row (power of two)
    constexpr int EPC = 16 / (int)sizeof(T);        // elements per chunk
    // thread -> (row, chunk); rows advance by nthr/CH per trip.  (gp, t) is tracked incrementally: no runtime
    // integer division inside the loop (Tn is a run-time value).
    const int c = threadIdx.x % CH, rstep = nthr / CH;
    int r = threadIdx.x / CH;
    int gp = r / Tn, t = r - gp * Tn;
    for (; r < GP * Tn; r += rstep) {
        int p = p0 + gp; if (p > P - 1) p = P - 1;
        const uint4 v = *reinterpret_cast<const uint4*>(src_base + ((n * Tn + t) * P + p) * ld + c * EPC);
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(lds_row<T>(dst, r, gp)) + c * 16) = v;
        t += rstep;
        while (t >= Tn) { t -= Tn; ++gp; }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_tm_fwd_lds(const T* __restrict__ qkv, T* __restrict__ o, float* __restrict__ lse,
                                                       int P, int H, int Tn, int GP, float scale) {
    MAED_DYN_SHARED(char, sm);
    const int C = H * D; const int64_t ld = 3 * (int64_t)C;
    const int chunks = (P + GP - 1) / GP;
    const int pc = blockIdx.x % chunks; int r0 = blockIdx.x / chunks;
    const int h = r0 % H; const int64_t n = r0 / H;
    const int p0 = pc * GP, nthr = GP * Tn;
    const size_t arr = (size_t)GP * Tn * RowGeom<T>::RS + GP * 16;
    char* Ks = sm; char* Vs = sm + arr;
    stage_group_rows<T>(Ks, qkv + C + h * D, ld, n, Tn, P, p0, GP, nthr);
    stage_group_rows<T>(Vs, qkv + 2 * C + h * D, ld, n, Tn, P, p0, GP, nthr);
    __syncthreads();
    const int gp = threadIdx.x / Tn, t = threadIdx.x % Tn;
    const int p = p0 + gp;
    if (p >= P) return;
    const int64_t f = n * Tn + t;
    float qv[D], acc[D];
    load_row(qkv + (f * P + p) * ld + h * D, qv);
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = 0.f;
    float m = -INFINITY, l = 0.f;
    for (int t2 = 0; t2 < Tn; ++t2) {
        const float s = dot_row(lds_row<T>(Ks, gp * Tn + t2, gp), qv) * scale;
        const float mn = fmaxf(m, s);
        const float a = __expf(m - mn), pr = __expf(s - mn);
        l = l * a + pr;
#pragma unroll
        for (int c = 0; c < D; ++c) acc[c] *= a;
        axpy_row(lds_row<T>(Vs, gp * Tn + t2, gp), pr, acc);
        m = mn;
    }
    store_row(o + (f * P + p) * C + h * D, acc, 1.f / l, 0);
    lse[(f * H + h) * P + p] = m + __logf(l);
}

// half-row helpers: two lanes (h2 = 0,1) share one (group, frame) row, each owning 32 of the 64 head dims, so the
// backward needs ~100 VGPRs instead of 256 (4 waves per SIMD instead of 1); dot products are completed with one
// lane-pair shuffle.
#define DH 32
template <typename T>
__device__ __forceinline__ void load_half(const T* p, float (&v)[DH]) {
#pragma unroll
    for (int c = 0; c < DH; c += 8) { float t[8]; ld8(p + c, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[c + j] = t[j]; }
}
template <typename T>
__device__ __forceinline__ float dot_half(const T* p, const float (&v)[DH]) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < DH; c += 8) { float t[8]; ld8(p + c, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) s = fmaf(v[c + j], t[j], s); }
    return s + __shfl_xor(s, 1, 64);
}
template <typename T>
__device__ __forceinline__ void axpy_half(const T* p, float a, float (&acc)[DH]) {
#pragma unroll
    for (int c = 0; c < DH; c += 8) { float t[8]; ld8(p + c, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[c + j] = fmaf(a, t[j], acc[c + j]); }
}
template <typename T>
__device__ __forceinline__ void store_half(T* p, const float (&v)[DH], int accumulate) {
#pragma unroll
    for (int c = 0; c < DH; c += 8) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = v[c + j];
        if (accumulate) { float o[8]; ld8(p + c, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] += o[j]; }
        st8(p + c, t);
    }
}

// blockDim = 256 = GP groups x Tn frames x 2 half-rows; tid = (gp*Tn + t)*2 + h2
template <typename T>
__global__ __launch_bounds__(256) void attn_tm_bwd_lds(const T* __restrict__ qkv, const T* __restrict__ o, const T* __restrict__ d_o,
                                                       const float* __restrict__ lse, T* __restrict__ dqkv, int accumulate,
                                                       int P, int H, int Tn, int GP, float scale) {
    MAED_DYN_SHARED(char, sm);
    const int C = H * D; const int64_t ld = 3 * (int64_t)C;
    const int chunks = (P + GP - 1) / GP;
    const int pc = blockIdx.x % chunks; int r0 = blockIdx.x / chunks;
    const int h = r0 % H; const int64_t n = r0 / H;
    const int p0 = pc * GP, nrows = GP * Tn;
    const size_t arr = (size_t)nrows * RowGeom<T>::RS + GP * 16;
    char* Qs = sm; char* Ks = sm + arr; char* Vs = sm + 2 * arr; char* dOs = sm + 3 * arr;
    float* Ls = reinterpret_cast<float*>(sm + 4 * arr); float* Ds = Ls + nrows;
    stage_group_rows<T>(Qs, qkv + h * D, ld, n, Tn, P, p0, GP, 256);
    stage_group_rows<T>(Ks, qkv + C + h * D, ld, n, Tn, P, p0, GP, 256);
    stage_group_rows<T>(Vs, qkv + 2 * C + h * D, ld, n, Tn, P, p0, GP, 256);
    stage_group_rows<T>(dOs, d_o + h * D, (int64_t)C, n, Tn, P, p0, GP, 256);
    const int h2 = threadIdx.x & 1, rt = threadIdx.x >> 1;     // half-row, row within the workgroup
    const int gp = rt / Tn, t = rt - gp * Tn;
    const int p = p0 + gp;
    const int pcl = p < P ? p : P - 1;
    const int64_t f = n * Tn + t;
    const int64_t row = f * P + pcl;
    const int g0 = gp * Tn, ho = h2 * DH;
    float a[DH], b[DH], acc[DH];
    load_half(d_o + row * C + h * D + ho, b);                           // dO (own half row)
    const float Dq = dot_half(o + row * C + h * D + ho, b);
    const float L = lse[(f * H + h) * P + pcl];
    if (h2 == 0) { Ds[rt] = Dq; Ls[rt] = L; }
    __syncthreads();
    if (p >= P) return;                                                 // both lanes of a pair leave together
    // ---- as query t: dQ = sum_t2 ds[t][t2] K[t2] ----
    load_half(lds_row<T>(Qs, g0 + t, gp) + ho, a);
#pragma unroll
    for (int c = 0; c < DH; ++c) acc[c] = 0.f;
    for (int t2 = 0; t2 < Tn; ++t2) {
        const T* kr = lds_row<T>(Ks, g0 + t2, gp) + ho;
        const float pr = __expf(dot_half(kr, a) * scale - L);
        const float ds = pr * (dot_half(lds_row<T>(Vs, g0 + t2, gp) + ho, b) - Dq) * scale;
        axpy_half(kr, ds, acc);
    }
    store_half(dqkv + row * ld + h * D + ho, acc, accumulate);
    // ---- as key t: dV = sum_t1 p[t1][t] dO[t1] ----
    load_half(lds_row<T>(Ks, g0 + t, gp) + ho, a);                     // k (own half row)
#pragma unroll
    for (int c = 0; c < DH; ++c) acc[c] = 0.f;
    for (int t1 = 0; t1 < Tn; ++t1) {
        const float pr = __expf(dot_half(lds_row<T>(Qs, g0 + t1, gp) + ho, a) * scale - Ls[g0 + t1]);
        axpy_half(lds_row<T>(dOs, g0 + t1, gp) + ho, pr, acc);
    }
    store_half(dqkv + row * ld + 2 * C + h * D + ho, acc, accumulate);
    // ---- as key t: dK = sum_t1 ds[t1][t] Q[t1] ----
    load_half(lds_row<T>(Vs, g0 + t, gp) + ho, b);                     // v (own half row)
#pragma unroll
    for (int c = 0; c < DH; ++c) acc[c] = 0.f;
    for (int t1 = 0; t1 < Tn; ++t1) {
        const T* q1 = lds_row<T>(Qs, g0 + t1, gp) + ho;
        const float pr = __expf(dot_half(q1, a) * scale - Ls[g0 + t1]);
        const float ds = pr * (dot_half(lds_row<T>(dOs, g0 + t1, gp) + ho, b) - Ds[g0 + t1]) * scale;
        axpy_half(q1, ds, acc);
    }
    store_half(dqkv + row * ld + C + h * D + ho, acc, accumulate);
}

template <typename T>
static bool launch_tm_fwd_lds(const void* qkv, void* o, float* lse, int F, int P, int H, int Tn, float scale, hipStream_t s) {
    if (Tn > 256 || 256 % Tn != 0) return false;
    const int GP = 256 / Tn;
    const size_t lds = 2 * ((size_t)256 * RowGeom<T>::RS + GP * 16);
    if (lds > 160 * 1024) return false;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)attn_tm_fwd_lds<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    const int chunks = (P + GP - 1) / GP;
    hipLaunchKernelGGL((attn_tm_fwd_lds<T>), dim3((unsigned)((F / Tn) * H * chunks)), dim3(256), lds, s, (const T*)qkv, (T*)o, lse, P, H, Tn, GP, scale);
    return true;
}
template <typename T>
static bool launch_tm_bwd_lds(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int accumulate, int F, int P,
                              int H, int Tn, float scale, hipStream_t s) {
    if (Tn > 128 || 128 % Tn != 0) return false;
    const int GP = 128 / Tn;
    const size_t lds = 4 * ((size_t)128 * RowGeom<T>::RS + GP * 16) + 2 * 128 * sizeof(float);
    if (lds > 160 * 1024) return false;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)attn_tm_bwd_lds<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    const int chunks = (P + GP - 1) / GP;
    hipLaunchKernelGGL((attn_tm_bwd_lds<T>), dim3((unsigned)((F / Tn) * H * chunks)), dim3(256), lds, s, (const T*)qkv, (const T*)o, (const T*)d_o,
                       lse, (T*)dqkv, accumulate, P, H, Tn, GP, scale);
    return true;
}

// ==================================================================================================
// MFMA forward (bf16): temporal attention as block-diagonal attention over a VIRTUAL sequence.
// A workgroup owns one (clip n, head h, group of G tokens): its L = G*T rows r = g*T + t are the T frames of G tokens
// (G = max(1, 32/T): with T = 16 two tokens fill one 32-row MFMA tile; with T = 64 one token is two tiles).  Row r lives at
// frame n*T + t, token tg*G + g of the (F,P,3C) qkv tensor.  S^T = K Q^T runs on v_mfma_f32_32x32x16_bf16 exactly as in the
// spatial kernel (lane = query, 16 keys per lane per tile); keys of a different token (or padding) are masked to -inf,
// the softmax is lane-local, O^T = V^T P^T consumes the exponentiated scores as the MFMA B fragment.  The thread-per-row
// kernels above spend ~2100 VALU instructions per query row on the dot products; this spends 8 MFMAs per 32x32 score tile.
// ==================================================================================================
__global__ __launch_bounds__(1024) void attn_tm_fwd_mfma(const bf16* __restrict__ qkv, bf16* __restrict__ o, float* __restrict__ lse,
                                                         int P, int H, int Tn, int G, int ngroups, float scale_log2e) {
    MAED_DYN_SHARED(unsigned short, smem);
    const int L = G * Tn, Lk = (L + 31) & ~31, VLD = Lk + 4;
    unsigned short* Ks = smem;                      // [Lk][64], 16-B chunk index XOR-swizzled with (row>>1)&7
    unsigned short* Vt = smem + (size_t)Lk * 64;    // [64][VLD]
    const int C = H * D;
    const int64_t ld = 3 * (int64_t)C;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    int bid = blockIdx.x;
    const int tg = bid % ngroups; bid /= ngroups;
    const int h = bid % H, n = bid / H;
    const int p0 = tg * G;
    // row r -> element offset of its (frame, token) row in qkv (q part, this head); rows beyond L or tokens beyond P are invalid
#define TM_ROW_OK(r) ((r) < L && p0 + (r) / Tn < P)
#define TM_ROW_OFF(r) ((((int64_t)n * Tn + (r) % Tn) * P + p0 + (r) / Tn) * ld + h * D)
    const int q = wave * 32 + l31;
    const bool q_ok = TM_ROW_OK(q);
    const int qg = q / Tn;
    bf16x8_t qf[4];
    {
        const int qc = q_ok ? q : 0;                // row 0 of the group always exists
        const bf16* qp = qkv + TM_ROW_OFF(qc);
#pragma unroll
        for (int t = 0; t < 4; ++t) qf[t] = *reinterpret_cast<const bf16x8_t*>(qp + t * 16 + hi * 8);
    }
    // stage K (swizzled rows) and V^T; invalid rows are staged as zeros (their scores are masked, 0 * finite = 0 in the PV product)
    for (int idx = tid; idx < Lk * 8; idx += nthr) {
        const int r = idx >> 3, c8 = (idx & 7) * 8;
        uint4 kv = make_uint4(0u, 0u, 0u, 0u), vv = kv;
        if (TM_ROW_OK(r)) {
            const bf16* rp = qkv + TM_ROW_OFF(r);
            kv = *reinterpret_cast<const uint4*>(rp + C + c8);
            vv = *reinterpret_cast<const uint4*>(rp + 2 * C + c8);
        }
        *reinterpret_cast<uint4*>(Ks + r * 64 + (((c8 >> 3) ^ ((r >> 1) & 7)) << 3)) = kv;
        const uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Vt[(c8 + 2 * j) * VLD + r] = (unsigned short)(w[j] & 0xffffu);
            Vt[(c8 + 2 * j + 1) * VLD + r] = (unsigned short)(w[j] >> 16);
        }
    }
    for (int i = tid; i < D * 4; i += nthr) Vt[(i >> 2) * VLD + Lk + (i & 3)] = 0;   // the 4 pad columns of every V^T row
    __syncthreads();

    f32x16_t oacc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; }
    float m = -INFINITY, l = 0.f;
    const int nkt = Lk / 32;
    for (int kt = 0; kt < nkt; ++kt) {
        f32x16_t s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const int krow = kt * 32 + l31;
        const unsigned short* kp = Ks + krow * 64;
        const int swz = (krow >> 1) & 7;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(kp + (((2 * t + hi) ^ swz) << 3));
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[t], s, 0, 0, 0);
        }
        // lane holds keys k(r) = kt*32 + (r&3) + 8*(r>>2) + 4*hi of its query: keep the keys of the query's own token
        bool any = false;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const bool ok = TM_ROW_OK(k) && (k / Tn == qg);
            s[r] = ok ? s[r] : -INFINITY;
            any |= ok;
        }
        float mt = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * scale_log2e;
        const float mn = fmaxf(m, mt);
        // a tile may hold no key of this query's token (mn stays -inf until the first one does): keep everything at zero then
        const float msafe = (mn == -INFINITY) ? 0.f : mn;
        const float alpha = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - msafe);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, -msafe)); ps += s[r]; }
        l = l * alpha + ps;
        m = mn;
#pragma unroll
        for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            union { bf16x8_t v; uint32_t u[4]; } pf;
#pragma unroll
            for (int j = 0; j < 4; ++j) pf.u[j] = pack_bf2(s[8 * st + 2 * j], s[8 * st + 2 * j + 1]);
#pragma unroll
            for (int et = 0; et < 2; ++et) {
                const unsigned short* vp = Vt + (et * 32 + l31) * VLD + kt * 32 + 16 * st + 4 * hi;
                union { bf16x8_t v; uint2 u[2]; } vf;
                vf.u[0] = *reinterpret_cast<const uint2*>(vp);
                vf.u[1] = *reinterpret_cast<const uint2*>(vp + 8);
                oacc[et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf.v, oacc[et], 0, 0, 0);
            }
        }
        (void)any;
    }
    l += __shfl_xor(l, 32, 64);
    if (Lk == 32) {      // one-wave workgroup (cfg3: T = 16): the output tile leaves as full 128-byte lines through the K image, whose last reader was this wave's S product
        const float inv = q_ok ? 1.f / l : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { oacc[0][r] *= inv; oacc[1][r] *= inv; }
        store_tile_lines_at(Ks, [&](int r) -> bf16* { return TM_ROW_OK(r) ? o + (((int64_t)n * Tn + r % Tn) * P + p0 + r / Tn) * C + h * D : nullptr; }, oacc, lane, 0);
        if (q_ok && hi == 0) lse[(((int64_t)n * Tn + q % Tn) * H + h) * P + p0 + qg] = (m + log2f(l)) * 0.69314718055994530942f;
    } else if (q_ok) {
        const float inv = 1.f / l;
        const int t = q % Tn, p = p0 + qg;
        const int64_t f = (int64_t)n * Tn + t;
        bf16* orow = o + (f * P + p) * C + h * D;
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int e0 = et * 32 + 8 * g + 4 * hi;
                const uint2 w = make_uint2(pack_bf2(oacc[et][4 * g] * inv, oacc[et][4 * g + 1] * inv),
                                           pack_bf2(oacc[et][4 * g + 2] * inv, oacc[et][4 * g + 3] * inv));
                *reinterpret_cast<uint2*>(orow + e0) = w;
            }
        if (hi == 0) lse[(f * H + h) * P + p] = (m + log2f(l)) * 0.69314718055994530942f;
    }
#undef TM_ROW_OK
#undef TM_ROW_OFF
}

// ==================================================================================================
// MFMA backward (bf16) over the same virtual sequences: ONE kernel produces dQ, dK and dV of a workgroup's L rows.
// Q, K, V, dO are staged once (row-major images for the S / dP products, transposed images for the dQ / dK / dV
// products), then every wave runs the query-side pass (lane = query: dS from S^T = K Q^T and dP^T = V dO^T, dQ^T += K^T dS^T)
// and the key-side pass (lane = key: P and dS from S = Q K^T, dV^T += dO^T P, dK^T += Q^T dS) for its own 32 rows --
// flash-style recompute from the saved log-sum-exp, block-diagonal mask as in the forward.
// ==================================================================================================
// Register budget: the workgroup is ONE wave at T = 16 and two at T = 64, so the common case is compiled for <= 256 threads at two
// workgroups per CU (166 VGPRs, nothing spilled); under the 1024-thread bound (128 VGPRs) the same code spills 40 VGPRs to scratch
// inside the loops.  The 1024-thread instantiation stays for virtual sequences longer than 128 rows.
template <int MAX_THREADS, int MIN_BLOCKS>
__global__ __launch_bounds__(MAX_THREADS, MIN_BLOCKS) void attn_tm_bwd_mfma(const bf16* __restrict__ qkv, const bf16* __restrict__ o, const bf16* __restrict__ d_o,
                                                         const float* __restrict__ lse, bf16* __restrict__ dqkv, int accumulate, int P, int H,
                                                         int Tn, int G, int ngroups, float scale) {
    MAED_DYN_SHARED(unsigned short, smem);
    const int L = G * Tn, Lk = (L + 31) & ~31, VLD = Lk + 4;
    unsigned short* Qs = smem;
    unsigned short* Ks = Qs + (size_t)Lk * KLD;
    unsigned short* Vs = Ks + (size_t)Lk * KLD;
    unsigned short* dOs = Vs + (size_t)Lk * KLD;
    unsigned short* Qt = dOs + (size_t)Lk * KLD;
    unsigned short* Kt = Qt + (size_t)D * VLD;
    unsigned short* dOt = Kt + (size_t)D * VLD;
    float* Ls = reinterpret_cast<float*>(dOt + (size_t)D * VLD);
    float* Ds = Ls + Lk;
    const int C = H * D;
    const int64_t ld = 3 * (int64_t)C;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    int bid = blockIdx.x;
    const int tg = bid % ngroups; bid /= ngroups;
    const int h = bid % H, n = bid / H;
    const int p0 = tg * G;
#define TM_ROW_OK(r) ((r) < L && p0 + (r) / Tn < P)
#define TM_TOK(r) (((int64_t)n * Tn + (r) % Tn) * P + p0 + (r) / Tn)          /* flattened (frame, token) index of row r */
    const float l2e = 1.44269504088896340736f;
    for (int idx = tid; idx < Lk * 8; idx += nthr) {
        const int r = idx >> 3, c8 = (idx & 7) * 8;
        uint4 qv = make_uint4(0u, 0u, 0u, 0u), kv = qv, vv = qv, gv = qv;
        if (TM_ROW_OK(r)) {
            const int64_t tok = TM_TOK(r);
            const bf16* rp = qkv + tok * ld + h * D;
            qv = *reinterpret_cast<const uint4*>(rp + c8);
            kv = *reinterpret_cast<const uint4*>(rp + C + c8);
            vv = *reinterpret_cast<const uint4*>(rp + 2 * C + c8);
            gv = *reinterpret_cast<const uint4*>(d_o + tok * C + h * D + c8);
        }
        *reinterpret_cast<uint4*>(Qs + r * KLD + c8) = qv;
        *reinterpret_cast<uint4*>(Ks + r * KLD + c8) = kv;
        *reinterpret_cast<uint4*>(Vs + r * KLD + c8) = vv;
        *reinterpret_cast<uint4*>(dOs + r * KLD + c8) = gv;
        const uint32_t wq[4] = {qv.x, qv.y, qv.z, qv.w}, wk[4] = {kv.x, kv.y, kv.z, kv.w}, wg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e0 = (c8 + 2 * j) * VLD + r, e1 = (c8 + 2 * j + 1) * VLD + r;
            Qt[e0] = (unsigned short)(wq[j] & 0xffffu); Qt[e1] = (unsigned short)(wq[j] >> 16);
            Kt[e0] = (unsigned short)(wk[j] & 0xffffu); Kt[e1] = (unsigned short)(wk[j] >> 16);
            dOt[e0] = (unsigned short)(wg[j] & 0xffffu); dOt[e1] = (unsigned short)(wg[j] >> 16);
        }
    }
    for (int i = tid; i < D * 4; i += nthr) {       // the 4 pad columns of the transposed images
        const int e = (i >> 2) * VLD + Lk + (i & 3);
        Qt[e] = 0; Kt[e] = 0; dOt[e] = 0;
    }
    for (int r = tid; r < Lk; r += nthr) {
        float dsum = 0.f, Lv = 0.f;
        if (TM_ROW_OK(r)) {
            const int64_t tok = TM_TOK(r);
#pragma unroll
            for (int c = 0; c < D; c += 8) {
                float a[8], b[8];
                ld8(d_o + tok * C + h * D + c, a); ld8(o + tok * C + h * D + c, b);
#pragma unroll
                for (int j = 0; j < 8; ++j) dsum = fmaf(a[j], b[j], dsum);
            }
            Lv = lse[(((int64_t)n * Tn + r % Tn) * H + h) * P + p0 + r / Tn] * l2e;
        }
        Ds[r] = dsum; Ls[r] = Lv;
    }
    __syncthreads();

    const int row = wave * 32 + l31;                // this lane's row: a query in pass A, a key in pass B
    const bool row_ok = TM_ROW_OK(row);
    const int rg = row / Tn;
    const float sl2e = scale * l2e;
    const int nt = Lk / 32;
    bf16* drow = dqkv + (row_ok ? TM_TOK(row) : 0) * ld + h * D;
    // ---- pass A: lane = query ------------------------------------------------------------------------------
    {
        bf16x8_t qf[4], dof[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            qf[t] = *reinterpret_cast<const bf16x8_t*>(Qs + row * KLD + t * 16 + hi * 8);
            dof[t] = *reinterpret_cast<const bf16x8_t*>(dOs + row * KLD + t * 16 + hi * 8);
        }
        const float Dq = Ds[row], L2 = Ls[row];
        f32x16_t dq[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }
        for (int kt = 0; kt < nt; ++kt) {
            f32x16_t sa, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = 0.f; }
            const int krow = kt * 32 + l31;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + krow * KLD + t * 16 + hi * 8);
                const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(Vs + krow * KLD + t * 16 + hi * 8);
                sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[t], sa, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[t], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const bool ok = row_ok && TM_ROW_OK(k) && (k / Tn == rg);
                const float pr = ok ? __builtin_amdgcn_exp2f(fmaf(sa[r], sl2e, -L2)) : 0.f;
                sa[r] = pr * (dp[r] - Dq) * scale;  // dS
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const bf16x8_t dsf = pack_frag(sa, st);
#pragma unroll
                for (int et = 0; et < 2; ++et) {
                    const bf16x8_t ktf = lds_frag_tr(Kt + (et * 32 + l31) * VLD + kt * 32 + 16 * st + 4 * hi);
                    dq[et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf, dq[et], 0, 0, 0);
                }
            }
        }
        if (row_ok) store_rowT(drow, dq, hi, accumulate);
    }
    // ---- pass B: lane = key ------------------------------------------------------------------------------------
    {
        bf16x8_t kf[4], vf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            kf[t] = *reinterpret_cast<const bf16x8_t*>(Ks + row * KLD + t * 16 + hi * 8);
            vf[t] = *reinterpret_cast<const bf16x8_t*>(Vs + row * KLD + t * 16 + hi * 8);
        }
        f32x16_t dk[2], dv[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }
        for (int qt = 0; qt < nt; ++qt) {
            f32x16_t sb, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sb[r] = 0.f; dp[r] = 0.f; }
            const int qrow = qt * 32 + l31;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8_t qfr = *reinterpret_cast<const bf16x8_t*>(Qs + qrow * KLD + t * 16 + hi * 8);
                const bf16x8_t dofr = *reinterpret_cast<const bf16x8_t*>(dOs + qrow * KLD + t * 16 + hi * 8);
                sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[t], sb, 0, 0, 0);     // D[q][k]: lane = key
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dofr, vf[t], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qq = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const bool ok = row_ok && TM_ROW_OK(qq) && (qq / Tn == rg);
                const float pr = ok ? __builtin_amdgcn_exp2f(fmaf(sb[r], sl2e, -Ls[qq])) : 0.f;
                dp[r] = pr * (dp[r] - Ds[qq]) * scale;  // dS
                sb[r] = pr;                             // P
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const bf16x8_t pf = pack_frag(sb, st), dsf = pack_frag(dp, st);
#pragma unroll
                for (int et = 0; et < 2; ++et) {
                    const int off = (et * 32 + l31) * VLD + qt * 32 + 16 * st + 4 * hi;
                    dv[et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag_tr(dOt + off), pf, dv[et], 0, 0, 0);
                    dk[et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag_tr(Qt + off), dsf, dk[et], 0, 0, 0);
                }
            }
        }
        if (row_ok) {
            store_rowT(drow + C, dk, hi, accumulate);
            store_rowT(drow + 2 * C, dv, hi, accumulate);
        }
    }
#undef TM_ROW_OK
#undef TM_TOK
}

// ==================================================================================================
// The same backward specialised for virtual sequences of ONE 32-row tile (T <= 32: cfg3's T = 16 packs two tokens per workgroup) --
// a one-wave workgroup.  With a single tile the "other side's" rows of both passes ARE the wave's own rows, so every row-major MFMA
// operand (Q, K, V, dO fragments) is loaded straight from global memory in fragment layout and only the A operands of the dQ / dK / dV
// products (Q^T, K^T, dO^T) go through LDS -- as row-major copies of the fragments the lane holds anyway, read back transposed by
// ds_read_b64_tr_b16: 13.8 KB per workgroup instead of 32.5 KB, i.e. 11 resident one-wave workgroups per CU instead of 4 (the general
// kernel runs at one wave per SIMD at T = 16).  Default for one-tile problems since it was timed on MI355X (profiles/r02_call1_attn_tm_wide_regs.txt: cfg3 backward 123.0 -> 73.8 us);
// MAED_TM_BWD_L32=0 switches back to the general kernel (A/B knob).
// ==================================================================================================
__global__ __launch_bounds__(64, 3) void attn_tm_bwd_mfma_l32(const bf16* __restrict__ qkv, const bf16* __restrict__ o, const bf16* __restrict__ d_o,
                                                               const float* __restrict__ lse, bf16* __restrict__ dqkv, int accumulate, int P, int H,
                                                               int Tn, int G, int ngroups, float scale) {
    // row-major images of the wave's own Q / K / dO rows: the transposed A operands of the dQ / dK / dV products are read from them with
    // ds_read_b64_tr_b16 (lds_frag_tr_rm) -- four 16-byte LDS writes per lane and operand instead of 32 two-byte writes of a transposed copy
    __shared__ __attribute__((aligned(16))) unsigned short Qs[32 * KLD], Ks[32 * KLD], dOs[32 * KLD];
    __shared__ float Ls[32], Ds[32];
    const int L = G * Tn, C = H * D;
    const int64_t ld = 3 * (int64_t)C;
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    int bid = blockIdx.x;
    const int tg = bid % ngroups; bid /= ngroups;
    const int h = bid % H, n = bid / H;
    const int p0 = tg * G;
#define TM_ROW_OK(r) ((r) < L && p0 + (r) / Tn < P)
#define TM_TOK(r) (((int64_t)n * Tn + (r) % Tn) * P + p0 + (r) / Tn)
    const int row = l31;
    const bool row_ok = TM_ROW_OK(row);
    const int64_t tok = row_ok ? TM_TOK(row) : 0;
    const bf16* rp = qkv + tok * ld + h * D;
    const bf16* gp = d_o + tok * C + h * D;
    const bf16* op = o + tok * C + h * D;
    const float l2e = 1.44269504088896340736f;
    // this lane's half of row `row`: elements t*16 + hi*8 .. +7, t = 0..3 (exactly the MFMA row-fragment layout)
    union Frag { bf16x8_t v; uint4 u; };
    Frag qf[4], kf[4], vf[4], dof[4];
    float dsum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int e0 = t * 16 + hi * 8;
        qf[t].u = kf[t].u = vf[t].u = dof[t].u = make_uint4(0u, 0u, 0u, 0u);
        if (row_ok) {
            qf[t].u = *reinterpret_cast<const uint4*>(rp + e0);
            kf[t].u = *reinterpret_cast<const uint4*>(rp + C + e0);
            vf[t].u = *reinterpret_cast<const uint4*>(rp + 2 * C + e0);
            dof[t].u = *reinterpret_cast<const uint4*>(gp + e0);
            float a[8], b[8];
            ld8(gp + e0, a); ld8(op + e0, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) dsum = fmaf(a[j], b[j], dsum);
        }
        *reinterpret_cast<uint4*>(Qs + row * KLD + e0) = qf[t].u;       // (rows beyond the sequence were zero-filled above)
        *reinterpret_cast<uint4*>(Ks + row * KLD + e0) = kf[t].u;
        *reinterpret_cast<uint4*>(dOs + row * KLD + e0) = dof[t].u;
    }
    dsum += __shfl_xor(dsum, 32, 64);
    const float Dq = dsum;
    const float L2 = row_ok ? lse[(((int64_t)n * Tn + row % Tn) * H + h) * P + p0 + row / Tn] * l2e : 0.f;
    if (hi == 0) { Ds[row] = Dq; Ls[row] = L2; }
    __syncthreads();

    const int rg = row / Tn;
    const float sl2e = scale * l2e;
    bf16* drow = dqkv + tok * ld + h * D;
    {   // ---- pass A: lane = query;  S^T = K Q^T, dP^T = V dO^T (rows = keys = the wave's own rows) ----
        f32x16_t sa, dp, dq[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = -Dq; dq[0][r] = 0.f; dq[1][r] = 0.f; }      // (dP - delta straight out of the accumulator)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[t].v, qf[t].v, sa, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[t].v, dof[t].v, dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const bool ok = row_ok && TM_ROW_OK(k) && (k / Tn == rg);
            const float pr = ok ? __builtin_amdgcn_exp2f(fmaf(sa[r], sl2e, -L2)) : 0.f;
            sa[r] = pr * dp[r];                      // dS / scale
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const bf16x8_t dsf = pack_frag(sa, st);
#pragma unroll
            for (int et = 0; et < 2; ++et)
                dq[et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag_tr_rm(Ks, 16 * st, et * 32, lane), dsf, dq[et], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { dq[0][r] *= scale; dq[1][r] *= scale; }
        // full 128-byte lines through the K image (its last reader was the dQ product above; one-wave workgroup): attn_mfma.cuh store_tile_lines_at
        store_tile_lines_at(Ks, [&](int r) -> bf16* { return TM_ROW_OK(r) ? dqkv + TM_TOK(r) * ld + h * D : nullptr; }, dq, lane, accumulate);
    }
    {   // ---- pass B: lane = key;  S = Q K^T, dP = dO V^T (rows = queries = the wave's own rows) ----
        f32x16_t sb, dp, dk[2], dv[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { sb[r] = 0.f; dp[r] = -Ds[(r & 3) + 8 * (r >> 2) + 4 * hi]; dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[t].v, kf[t].v, sb, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof[t].v, vf[t].v, dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const bool ok = row_ok && TM_ROW_OK(qq) && (qq / Tn == rg);
            const float pr = ok ? __builtin_amdgcn_exp2f(fmaf(sb[r], sl2e, -Ls[qq])) : 0.f;
            dp[r] *= pr;                             // dS / scale
            sb[r] = pr;                              // P
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const bf16x8_t pf = pack_frag(sb, st), dsf = pack_frag(dp, st);
#pragma unroll
            for (int et = 0; et < 2; ++et) {
                dv[et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag_tr_rm(dOs, 16 * st, et * 32, lane), pf, dv[et], 0, 0, 0);
                dk[et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag_tr_rm(Qs, 16 * st, et * 32, lane), dsf, dk[et], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[0][r] *= scale; dk[1][r] *= scale; }
        store_tile_lines_at(Ks, [&](int r) -> bf16* { return TM_ROW_OK(r) ? dqkv + TM_TOK(r) * ld + h * D + C : nullptr; }, dk, lane, accumulate);
        store_tile_lines_at(Ks, [&](int r) -> bf16* { return TM_ROW_OK(r) ? dqkv + TM_TOK(r) * ld + h * D + 2 * C : nullptr; }, dv, lane, accumulate);
    }
#undef TM_ROW_OK
#undef TM_TOK
}

// bf16: always the MFMA kernels.  Measured on MI355X against the LDS-staged thread-per-row kernels, which remain the f32 parity path
// (profiles/r01_attn_temporal_mfma_ab.txt): cfg3 (T=16) forward 61.8 -> 27.7 us, backward 133.8 -> 123.3 us; cfg5 (T=64) forward 251 -> 62 us,
// backward 578 -> 304 us.

static bool launch_tm_bwd_mfma(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int accumulate, int F, int P, int H,
                               int Tn, float scale, hipStream_t s) {
    const int G = Tn >= 32 ? 1 : 32 / Tn;
    const int L = G * Tn, Lk = (L + 31) & ~31;
    if (Lk / 32 > 16) return false;
    const size_t lds = ((size_t)4 * Lk * KLD + (size_t)3 * D * (Lk + 4)) * 2 + (size_t)2 * Lk * sizeof(float);
    if (lds > 160 * 1024) return false;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)attn_tm_bwd_mfma<256, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)attn_tm_bwd_mfma<1024, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    const int ngroups = (P + G - 1) / G;
    const dim3 grid((unsigned)((F / Tn) * H * ngroups)), block(64 * (Lk / 32));
    if (Lk == 32) {                                                   // one-tile specialisation (see above)
        hipLaunchKernelGGL(attn_tm_bwd_mfma_l32, grid, dim3(64), 0, s, (const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse, (bf16*)dqkv, accumulate,
                           P, H, Tn, G, ngroups, scale);
        return true;
    }
    // up to four waves per workgroup: the spill-free instantiation (166 VGPRs; measured on MI355X, profiles/r02_call1_attn_tm_wide_regs.txt:
    // cfg5 T = 64 backward 305.8 -> 215.9 us, cfg3 123.0 -> 90.6 us)
    if (Lk / 32 <= 4)
        hipLaunchKernelGGL((attn_tm_bwd_mfma<256, 2>), grid, block, lds, s, (const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse, (bf16*)dqkv,
                           accumulate, P, H, Tn, G, ngroups, scale);
    else
        hipLaunchKernelGGL((attn_tm_bwd_mfma<1024, 1>), grid, block, lds, s, (const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse, (bf16*)dqkv,
                           accumulate, P, H, Tn, G, ngroups, scale);
    return true;
}

static bool launch_tm_fwd_mfma(const void* qkv, void* o, float* lse, int F, int P, int H, int Tn, float scale, hipStream_t s) {
    const int G = Tn >= 32 ? 1 : 32 / Tn;
    const int L = G * Tn, Lk = (L + 31) & ~31;
    if (Lk / 32 > 16) return false;                                   // at most 16 waves per workgroup
    const size_t lds = ((size_t)Lk * 64 + (size_t)D * (Lk + 4)) * 2;
    if (lds > 160 * 1024) return false;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)attn_tm_fwd_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    const int ngroups = (P + G - 1) / G;
    hipLaunchKernelGGL(attn_tm_fwd_mfma, dim3((unsigned)((F / Tn) * H * ngroups)), dim3(64 * (Lk / 32)), lds, s, (const bf16*)qkv, (bf16*)o, lse, P, H,
                       Tn, G, ngroups, scale * 1.44269504088896340736f);
    return true;
}

extern "C" int maed_attn_temporal_fwd(const void* qkv, void* o, float* lse, int F, int P, int H, int T, float scale, int dtype,
                                      void* stream) {
    MAED_CHECK_ARG(qkv && o && lse, MAED_ERR_ARG, "attn_temporal_fwd: null pointer");
    MAED_CHECK_ARG(T > 0 && F % T == 0 && P > 0 && H > 0, MAED_ERR_SHAPE, "attn_temporal_fwd: F=%d must be a multiple of T=%d", F, T);
    MAED_CHECK_ARG(is_aligned(qkv, 16) && is_aligned(o, 16), MAED_ERR_ALIGN, "attn_temporal_fwd: alignment");
    const int64_t total = (int64_t)F * H * P;
    if (total == 0) return MAED_OK;
    dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == MAED_BF16 && launch_tm_fwd_mfma(qkv, o, lse, F, P, H, T, scale, (hipStream_t)stream)) { MAED_CHECK_LAUNCH("attn_temporal_fwd"); return MAED_OK; }
    // fp32 operands, split-bf16 contractions on the matrix cores (attn_x3.hip): MAED_F32X3 per call or the process-wide fp32 matmul mode; bf16x6 keeps the exact kernels
    const int np_call = maed_x3_take_dtype(dtype);
    if (dtype == MAED_F32 && maed_attn_tm_x3_fwd_launch(np_call ? np_call : maed_x3_planes(), qkv, o, lse, F, P, H, T, scale, (hipStream_t)stream)) {
        MAED_CHECK_LAUNCH("attn_temporal_fwd(x3)");
        return MAED_OK;
    }
    MAED_DISPATCH_DTYPE(dtype, TT, {
        if (!launch_tm_fwd_lds<TT>(qkv, o, lse, F, P, H, T, scale, (hipStream_t)stream))
            hipLaunchKernelGGL((attn_tm_fwd_kernel<TT>), grid, dim3(256), 0, (hipStream_t)stream, (const TT*)qkv, (TT*)o, lse, total, P, H, T, scale);
    });
    MAED_CHECK_LAUNCH("attn_temporal_fwd");
    return MAED_OK;
}

extern "C" int maed_attn_temporal_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv,
                                      int accumulate, int F, int P, int H, int T, float scale, int dtype, void* stream) {
    MAED_CHECK_ARG(qkv && o && d_o && lse && dqkv, MAED_ERR_ARG, "attn_temporal_bwd: null pointer");
    MAED_CHECK_ARG(T > 0 && F % T == 0 && P > 0 && H > 0, MAED_ERR_SHAPE, "attn_temporal_bwd: F=%d must be a multiple of T=%d", F, T);
    const int64_t total = (int64_t)F * H * P;
    if (total == 0) return MAED_OK;
    dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == MAED_BF16 && launch_tm_bwd_mfma(qkv, o, d_o, lse, dqkv, accumulate, F, P, H, T, scale, (hipStream_t)stream)) {
        MAED_CHECK_LAUNCH("attn_temporal_bwd");
        return MAED_OK;
    }
    const int np_call = maed_x3_take_dtype(dtype);
    if (dtype == MAED_F32 && maed_attn_tm_x3_bwd_launch(np_call ? np_call : maed_x3_planes(), qkv, o, d_o, lse, dqkv, accumulate, F, P, H, T, scale, (hipStream_t)stream)) {
        MAED_CHECK_LAUNCH("attn_temporal_bwd(x3)");
        return MAED_OK;
    }
    MAED_DISPATCH_DTYPE(dtype, TT, {
        if (!launch_tm_bwd_lds<TT>(qkv, o, d_o, lse, dqkv, accumulate, F, P, H, T, scale, (hipStream_t)stream))
            hipLaunchKernelGGL((attn_tm_bwd_kernel<TT>), grid, dim3(256), 0, (hipStream_t)stream, (const TT*)qkv, (const TT*)o, (const TT*)d_o, lse,
                               (TT*)dqkv, accumulate, total, P, H, T, scale);
    });
    MAED_CHECK_LAUNCH("attn_temporal_bwd");
    return MAED_OK;
}
