// K4: temporal attention of the STE (vision_transformer.py:216-228): per (clip n, head h, token p)
//     attention across the T frames of the clip.  The reference makes three permuted copies of
//     q/k/v and one of the output per block; here every thread gathers its rows straight from the
//     (F,P,3C) qkv buffer (frame stride P*3C) and writes the (F,P,C) result in place.
// HBM-bound (arithmetic intensity ~T/2 flop/B): thread per (n,h,p,t) query row, the T key/value rows
// of the same (n,h,p) are shared by T neighbouring threads through L1.  fp32 math for both dtypes.
#include "common.cuh"

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
    if (!attr) { hipFuncSetAttribute((const void*)attn_tm_fwd_lds<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
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
    if (!attr) { hipFuncSetAttribute((const void*)attn_tm_bwd_lds<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    const int chunks = (P + GP - 1) / GP;
    hipLaunchKernelGGL((attn_tm_bwd_lds<T>), dim3((unsigned)((F / Tn) * H * chunks)), dim3(256), lds, s, (const T*)qkv, (const T*)o, (const T*)d_o,
                       lse, (T*)dqkv, accumulate, P, H, Tn, GP, scale);
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
    MAED_DISPATCH_DTYPE(dtype, TT, {
        if (!launch_tm_bwd_lds<TT>(qkv, o, d_o, lse, dqkv, accumulate, F, P, H, T, scale, (hipStream_t)stream))
            hipLaunchKernelGGL((attn_tm_bwd_kernel<TT>), grid, dim3(256), 0, (hipStream_t)stream, (const TT*)qkv, (const TT*)o, (const TT*)d_o, lse,
                               (TT*)dqkv, accumulate, total, P, H, T, scale);
    });
    MAED_CHECK_LAUNCH("attn_temporal_bwd");
    return MAED_OK;
}
