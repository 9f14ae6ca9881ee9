// Backbone-side HBM-bound helpers.  The convolutions of the hybrid R50 ride on MIOpen (north_star); what
// MIOpen does not give us is fused here:
//   * weight standardisation of ALL StdConv2dSame filters (resnetv2.py:74-93) in ONE launch forward and ONE
//     launch backward (the reference recomputes it twice per conv call through ~6 ATen ops each; per step
//     that was ~1000 tiny launches), writing the compute-dtype weights directly in MIOpen's channels_last
//     (O, kh, kw, I) order;
//   * GroupNorm(32) + ReLU (resnetv2.py:35-49) on channels_last activations: statistics pass + fused
//     normalise/affine/ReLU pass forward; backward as one reduction pass + one apply pass with the ReLU
//     mask recomputed (ATen: three passes plus separate ReLU forward/backward passes).
#include "common.cuh"

// ---------------------------------------------------------------------------------------------------------
// weight standardisation, batched over all convolutions of the backbone
// ---------------------------------------------------------------------------------------------------------
struct WsConv {           // one per convolution (host builds the table; <= 64 entries)
    const float* w;       // fp32 master weight (O, I, kh, kw) contiguous
    float* gw;            // fp32 gradient of w (same layout), += target            (backward only)
    const void* gout;     // gradient w.r.t. the standardised weight, (O, kh, kw, I) (backward only)
    int64_t dst_off;      // element offset of this conv in the standardised-weight arena, (O, kh, kw, I)
    int64_t dst_t_off;    // >= 0: also write the TRANSPOSED (kh*kw*I, O) image there (operand of the input-gradient GEMM of a 1x1 conv)
    int32_t O, I, KHW, fstart;  // fstart = index of this conv's first filter in the global filter numbering
    int32_t gout_f32, pad_;     // gout is fp32 (written by maed_gemm_tn_wgrad) instead of the compute dtype
};
#define WS_MAX_CONVS 64

__device__ __forceinline__ int ws_find_conv(const WsConv* __restrict__ tab, int n_convs, int fi) {
    int c = 0;
    for (int i = 1; i < n_convs; ++i) c = (fi >= tab[i].fstart) ? i : c;   // wave-uniform, table is tiny
    return c;
}

// one wave per filter
template <typename T>
__global__ __launch_bounds__(256) void ws_fwd_kernel(const WsConv* __restrict__ tab, int n_convs, int n_filters, T* __restrict__ out,
                                                     float* __restrict__ stats /* [n_filters][2]: mean, 1/(std+eps) */, float eps, int skip_transposed) {
    const int fi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (fi >= n_filters) return;
    const WsConv d = tab[ws_find_conv(tab, n_convs, fi)];
    const int K = d.I * d.KHW, o = fi - d.fstart;
    const float* src = d.w + (int64_t)o * K;
    float s = 0.f;
    for (int i = lane; i < K; i += 64) s += src[i];
    const float mean = wave_sum(s) / (float)K;
    float q = 0.f;
    for (int i = lane; i < K; i += 64) { const float c = src[i] - mean; q += c * c; }
    const float inv = 1.f / (sqrtf(wave_sum(q) / (float)K) + eps);   // biased std, eps added to the std (resnetv2.py:87-88)
    if (lane == 0) { stats[2 * fi] = mean; stats[2 * fi + 1] = inv; }
    T* dst = out + d.dst_off + (int64_t)o * K;
    T* dst_t = (d.dst_t_off >= 0 && !skip_transposed) ? out + d.dst_t_off + o : nullptr;      // (skip: ws_transpose_kernel writes them)
    // source (ci, r), destination (r, ci): lane = input channel, taps in the inner loop -- the stores of one tap are consecutive elements (whole lines for the 3x3
    // filters, 63 % of the weights; with lane = source index they were 2-byte pieces nine rows apart, and every element paid an integer division by KHW)
    const int KHW = d.KHW, I = d.I;
    for (int ci = lane; ci < I; ci += 64) {
        const float* sp = src + (int64_t)ci * KHW;
        for (int r = 0; r < KHW; ++r) {
            const float v = (sp[r] - mean) * inv;
            stf(dst + (int64_t)r * I + ci, v);
            if (dst_t) stf(dst_t + ((int64_t)r * I + ci) * d.O, v);
        }
    }
}

// Round 4: the TRANSPOSED images by a tile transpose of the (O, K) image the kernel above has just written, instead of element-wise stores with a stride of O
// between lanes (2-byte stores into 2-byte-wide columns: 11.9 M scattered stores per step at cfg3; the wave-per-filter kernel took 156 us for 94 MB of
// weights).  One workgroup per 64 x 64 tile of one convolution (the tile -> convolution map is a scan over the <= 64-entry table), 128-byte runs both ways.
// (A first version that also moved the statistics into 64-filter workgroups -- 16 filters per wave, one after the other -- was SLOWER than the original:
// 21.49 vs 21.14 ms per step; the statistics are latency-bound and want the 26 K independent waves of the kernel above.)
template <typename T>
__global__ __launch_bounds__(256) void ws_transpose_kernel(const WsConv* __restrict__ tab, int n_convs, T* __restrict__ out) {
    __shared__ T tile[64][66];
    int t = blockIdx.x, c = 0, tk = 0;
    for (; c < n_convs; ++c) {                       // block-uniform scan: which convolution owns tile t
        const int K = tab[c].I * tab[c].KHW;
        tk = (K + 63) / 64;
        const int nt = tab[c].dst_t_off >= 0 ? ((tab[c].O + 63) / 64) * tk : 0;
        if (t < nt) break;
        t -= nt;
    }
    if (c >= n_convs) return;
    const WsConv d = tab[c];
    const int K = d.I * d.KHW, o0 = (t / tk) * 64, k0 = (t % tk) * 64;
    const T* src = out + d.dst_off;                  // (O, K)
    T* dst = out + d.dst_t_off;                      // (K, O)
    const int r = threadIdx.x >> 2, q = (threadIdx.x & 3) * 16;
    // whole 16-element runs as two 16-byte accesses (K and O are multiples of 8 for every convolution that has a transposed image, the arenas 16-byte aligned:
    // checked per run, scalar otherwise)
    constexpr int VE = 16 / (int)sizeof(T);          // elements per 16 bytes
    if (o0 + r < d.O) {
        const T* sp = src + (int64_t)(o0 + r) * K + k0 + q;
        if (k0 + q + 16 <= K && ((uintptr_t)sp & 15) == 0) {
#pragma unroll
            for (int v = 0; v < 16 / VE; ++v) { const uint4 w = reinterpret_cast<const uint4*>(sp)[v]; T e[VE]; memcpy(e, &w, 16);
#pragma unroll
                for (int j = 0; j < VE; ++j) tile[r][q + v * VE + j] = e[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) if (k0 + q + j < K) tile[r][q + j] = sp[j];
        }
    }
    __syncthreads();
    if (k0 + r < K) {
        T* dp = dst + (int64_t)(k0 + r) * d.O + o0 + q;
        if (o0 + q + 16 <= d.O && ((uintptr_t)dp & 15) == 0) {
#pragma unroll
            for (int v = 0; v < 16 / VE; ++v) { T e[VE];
#pragma unroll
                for (int j = 0; j < VE; ++j) e[j] = tile[q + v * VE + j][r];
                uint4 w; memcpy(&w, e, 16); reinterpret_cast<uint4*>(dp)[v] = w; }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) if (o0 + q + j < d.O) dp[j] = tile[q + j][r];
        }
    }
}

// dw_i += inv*(g_i - mean(g)) - (w_i - mu) * inv^2 / (K*sigma) * sum_j g_j (w_j - mu),   sigma = 1/inv - eps
template <typename T>
__global__ __launch_bounds__(256) void ws_bwd_kernel(const WsConv* __restrict__ tab, int n_convs, int n_filters,
                                                     const float* __restrict__ stats, float eps) {
    const int fi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (fi >= n_filters) return;
    const WsConv d = tab[ws_find_conv(tab, n_convs, fi)];
    if (!d.gout || !d.gw) return;   // this convolution received no gradient
    const int K = d.I * d.KHW, o = fi - d.fstart;
    const float* src = d.w + (int64_t)o * K;
    const T* g = (const T*)d.gout + (int64_t)o * K;
    const float* g32 = (const float*)d.gout + (int64_t)o * K;
    const bool f32 = d.gout_f32 != 0;               // wave-uniform
    const float mean = stats[2 * fi], inv = stats[2 * fi + 1];
    // lane = SOURCE index here (unlike ws_fwd_kernel): the read-modify-write of the fp32 gradient is the side that has to be whole lines -- with lane = input channel it
    // became 4-byte pieces nine elements apart and the kernel went from 87 to 119 us; the gather of the (r, ci)-ordered incoming gradient is absorbed by the caches.
    // ci = i / KHW by a float reciprocal (exact for these sizes: i < 2^20, KHW in {1, 9, 49}) instead of an integer division per element.
    const int KHW = d.KHW, I = d.I;
    const float rk = 1.0f / (float)KHW;
    float sg = 0.f, sgw = 0.f;
    for (int i = lane; i < K; i += 64) {
        const int ci = KHW == 1 ? i : (int)(((float)i + 0.5f) * rk), r = i - ci * KHW;
        const float gi = f32 ? g32[(int64_t)r * I + ci] : ldf(g + (int64_t)r * I + ci);
        sg += gi; sgw += gi * (src[i] - mean);
    }
    sg = wave_sum(sg); sgw = wave_sum(sgw);
    const float sigma = 1.f / inv - eps;
    const float c2 = (sigma > 0.f) ? sgw * inv * inv / ((float)K * sigma) : 0.f;
    const float gm = sg / (float)K;
    float* dst = d.gw + (int64_t)o * K;
    for (int i = lane; i < K; i += 64) {
        const int ci = KHW == 1 ? i : (int)(((float)i + 0.5f) * rk), r = i - ci * KHW;
        const float gi = f32 ? g32[(int64_t)r * I + ci] : ldf(g + (int64_t)r * I + ci);
        dst[i] += inv * (gi - gm) - (src[i] - mean) * c2;
    }
}

extern "C" int maed_weight_std_fwd(const void* conv_table, int n_convs, int n_filters, void* out, int dtype, float* stats, float eps,
                                   int filters_aligned64, void* stream) {
    MAED_CHECK_ARG(conv_table && out && stats, MAED_ERR_ARG, "weight_std_fwd: null pointer");
    MAED_CHECK_ARG(n_convs > 0 && n_convs <= WS_MAX_CONVS, MAED_ERR_SHAPE, "weight_std_fwd: n_convs=%d out of range", n_convs);
    if (n_filters <= 0) return MAED_OK;
    // filters_aligned64 > 0: the transposed images by ws_transpose_kernel; its value is the number of 64 x 64 tiles of all convolutions that have one
    // (the table lives in device memory: the caller counts)
    const int t_tiles = filters_aligned64;
    MAED_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL((ws_fwd_kernel<T>), dim3((n_filters + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const WsConv*)conv_table, n_convs, n_filters, (T*)out, stats, eps,
                           t_tiles > 0 ? 1 : 0);
        if (t_tiles > 0) hipLaunchKernelGGL((ws_transpose_kernel<T>), dim3(t_tiles), dim3(256), 0, (hipStream_t)stream, (const WsConv*)conv_table, n_convs, (T*)out);
    });
    MAED_CHECK_LAUNCH("weight_std_fwd");
    return MAED_OK;
}

extern "C" int maed_weight_std_bwd(const void* conv_table, int n_convs, int n_filters, int dtype, const float* stats, float eps, void* stream) {
    MAED_CHECK_ARG(conv_table && stats, MAED_ERR_ARG, "weight_std_bwd: null pointer");
    MAED_CHECK_ARG(n_convs > 0 && n_convs <= WS_MAX_CONVS, MAED_ERR_SHAPE, "weight_std_bwd: n_convs=%d out of range", n_convs);
    if (n_filters <= 0) return MAED_OK;
    MAED_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((ws_bwd_kernel<T>), dim3((n_filters + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                                                      (const WsConv*)conv_table, n_convs, n_filters, stats, eps));
    MAED_CHECK_LAUNCH("weight_std_bwd");
    return MAED_OK;
}

// ---------------------------------------------------------------------------------------------------------
// GroupNorm(32 groups) [+ residual add] [+ ReLU] on channels_last activations x[n][hw][c]
// ---------------------------------------------------------------------------------------------------------
#define GN_G 32

// a thread owns 8 consecutive channels (16 B of bf16); threads of a workgroup tile (rows x channel blocks)
template <typename T>
__device__ __forceinline__ void gn_load8(const T* p, float (&v)[8]) { ld8(p, v); }

// ---- forward pass 1: per (n, group) sums in double (fp32 partials per thread, double across threads) -------
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, double* __restrict__ sums /* [N][32][2] */, int HW, int C,
                                                       int rows_per_wg) {
    __shared__ double ls[GN_G][2];
    const int n = blockIdx.y, cbn = C / 8, cb = threadIdx.x % cbn, rsub = threadIdx.x / cbn, rstep = 256 / cbn;
    if (threadIdx.x < GN_G * 2) (&ls[0][0])[threadIdx.x] = 0.0;
    __syncthreads();
    const int r0 = blockIdx.x * rows_per_wg;
    const int r1 = min(HW, r0 + rows_per_wg);
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
    const T* base = x + ((int64_t)n * HW) * C + cb * 8;
    for (int r = r0 + rsub; r < r1; r += rstep) {
        float v[8];
        gn_load8(base + (int64_t)r * C, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] += v[j]; q[j] = fmaf(v[j], v[j], q[j]); }
    }
    const int cpg = C / GN_G;
    if (cpg >= 8) {
        float ss = 0.f, qq = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { ss += s[j]; qq += q[j]; }
        const int g = cb * 8 / cpg;
        atomicAdd(&ls[g][0], (double)ss); atomicAdd(&ls[g][1], (double)qq);
    } else {
        for (int j0 = 0; j0 < 8; j0 += cpg) {
            float ss = 0.f, qq = 0.f;
            for (int j = j0; j < j0 + cpg; ++j) { ss += s[j]; qq += q[j]; }
            const int g = (cb * 8 + j0) / cpg;
            atomicAdd(&ls[g][0], (double)ss); atomicAdd(&ls[g][1], (double)qq);
        }
    }
    __syncthreads();
    if (threadIdx.x < GN_G * 2) atomicAdd(sums + (int64_t)n * GN_G * 2 + threadIdx.x, (&ls[0][0])[threadIdx.x]);
}


// ---- forward pass 2: y = act((x - mu) * rstd * gamma + beta [+ residual]) ------------------------------------
template <typename T, bool RES, bool RELU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const T* __restrict__ res, const double* __restrict__ sums,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y,
                                                       uint8_t* __restrict__ mask, int HW, int C, float eps, int rows_per_wg,
                                                       bf16* __restrict__ twin_x = nullptr, bf16* __restrict__ twin_y = nullptr) {
    // twin_x / twin_y (maed_groupnorm_fwd_twin, T = float): bf16 copies of the input and of the result for a bf16 backward, written from the registers of this pass
    __shared__ float lmu[GN_G], lrs[GN_G];
    const int n = blockIdx.y;
    if (threadIdx.x < GN_G) {
        const double cnt = (double)HW * (C / GN_G);
        const double m = sums[((int64_t)n * GN_G + threadIdx.x) * 2] / cnt;
        double var = sums[((int64_t)n * GN_G + threadIdx.x) * 2 + 1] / cnt - m * m;
        if (var < 0.0) var = 0.0;
        lmu[threadIdx.x] = (float)m; lrs[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int cbn = C / 8, cb = threadIdx.x % cbn, rsub = threadIdx.x / cbn, rstep = 256 / cbn, cpg = C / GN_G;
    float a[8], b[8];   // y = x*a + b
    {
        float gg[8], bb[8];
        ld8(gamma + cb * 8, gg); ld8(beta + cb * 8, bb);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int g = (cb * 8 + j) / cpg;
            a[j] = lrs[g] * gg[j]; b[j] = bb[j] - lmu[g] * a[j];
        }
    }
    const int r0 = blockIdx.x * rows_per_wg, r1 = min(HW, r0 + rows_per_wg);
    const int64_t base = ((int64_t)n * HW) * C + cb * 8;
    for (int r = r0 + rsub; r < r1; r += rstep) {
        float v[8], o[8];
        gn_load8(x + base + (int64_t)r * C, v);
        if (twin_x) st8_nt(twin_x + base + (int64_t)r * C, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaf(v[j], a[j], b[j]);
        if (RES) { float rr[8]; gn_load8(res + base + (int64_t)r * C, rr);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += rr[j]; }
        if (RELU) {
            if (RES && mask) {   // 1 bit per element for the backward (the ReLU mask cannot be recomputed from x once a residual was added)
                uint32_t bits = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) bits |= (o[j] > 0.f ? 1u : 0u) << j;
                mask[((int64_t)n * HW + r) * cbn + cb] = (uint8_t)bits;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = fmaxf(o[j], 0.f); }
        st8(y + base + (int64_t)r * C, o);
        if (twin_y) st8_nt(twin_y + base + (int64_t)r * C, o);
    }
}

// ---- backward pass 1: ab[n][c] = (sum_hw dy_eff, sum_hw dy_eff * xhat); dgamma/dbeta accumulated here too ----
// dy_eff = dy * (out > 0) when RELU; the mask is recomputed from x when there is no residual (YMASK=false), read from the forward's
// 1-bit-per-element mask otherwise (16x less traffic than re-reading the saved output)
template <typename T, bool RELU, bool YMASK>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const T* __restrict__ x, const uint8_t* __restrict__ mask, const T* __restrict__ dy,
                                                            const double* __restrict__ sums, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ ab /* [N][C][2] */,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int HW, int C, float eps, int rows_per_wg, int defer_affine) {
    MAED_DYN_SHARED(float, lpart);   // [256/cbn][C][2] = 16 KB
    __shared__ float lmu[GN_G], lrs[GN_G];
    const int n = blockIdx.y;
    if (threadIdx.x < GN_G) {
        const double cnt = (double)HW * (C / GN_G);
        const double m = sums[((int64_t)n * GN_G + threadIdx.x) * 2] / cnt;
        double var = sums[((int64_t)n * GN_G + threadIdx.x) * 2 + 1] / cnt - m * m;
        if (var < 0.0) var = 0.0;
        lmu[threadIdx.x] = (float)m; lrs[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int cbn = C / 8, cb = threadIdx.x % cbn, rsub = threadIdx.x / cbn, rstep = 256 / cbn, cpg = C / GN_G;
    float mu[8], rs[8], sa[8], sb[8], ga[8], be[8];
    if (RELU && !YMASK) { ld8(gamma + cb * 8, ga); ld8(beta + cb * 8, be); }
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int g = (cb * 8 + j) / cpg; mu[j] = lmu[g]; rs[j] = lrs[g]; sa[j] = 0.f; sb[j] = 0.f; }
    const int r0 = blockIdx.x * rows_per_wg, r1 = min(HW, r0 + rows_per_wg);
    const int64_t base = ((int64_t)n * HW) * C + cb * 8;
    // four rows per trip: 8-12 independent 16-B loads in flight per thread (this pass runs on ~3 workgroups per CU to keep
    // the closing atomics few, so the memory-level parallelism has to come from inside the thread)
#define GN_RED_ROW(v_, d_, o_)                                                                                  \
    {                                                                                                           \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) v_[j] = (v_[j] - mu[j]) * rs[j];   /* xhat */            \
        if (RELU) {                                                                                             \
            if (YMASK) { _Pragma("unroll") for (int j = 0; j < 8; ++j) d_[j] = ((o_ >> j) & 1u) ? d_[j] : 0.f; }  \
            else { _Pragma("unroll") for (int j = 0; j < 8; ++j) d_[j] = fmaf(v_[j], ga[j], be[j]) > 0.f ? d_[j] : 0.f; } \
        }                                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) { sa[j] += d_[j]; sb[j] = fmaf(d_[j], v_[j], sb[j]); }    \
    }
    int r = r0 + rsub;
    for (; r + 3 * rstep < r1; r += 4 * rstep) {
        float v0[8], v1[8], v2[8], v3[8], d0[8], d1[8], d2[8], d3[8];
        uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0;
        const int64_t a0 = base + (int64_t)r * C, a1 = a0 + (int64_t)rstep * C, a2 = a1 + (int64_t)rstep * C, a3 = a2 + (int64_t)rstep * C;
        gn_load8(x + a0, v0); gn_load8(x + a1, v1); gn_load8(x + a2, v2); gn_load8(x + a3, v3);
        gn_load8(dy + a0, d0); gn_load8(dy + a1, d1); gn_load8(dy + a2, d2); gn_load8(dy + a3, d3);
        if (RELU && YMASK) {
            const int64_t m0 = ((int64_t)n * HW + r) * cbn + cb;
            o0 = mask[m0]; o1 = mask[m0 + (int64_t)rstep * cbn]; o2 = mask[m0 + 2 * (int64_t)rstep * cbn]; o3 = mask[m0 + 3 * (int64_t)rstep * cbn];
        }
        GN_RED_ROW(v0, d0, o0) GN_RED_ROW(v1, d1, o1) GN_RED_ROW(v2, d2, o2) GN_RED_ROW(v3, d3, o3)
    }
    for (; r < r1; r += rstep) {
        float v[8], d[8];
        uint32_t o = 0;
        gn_load8(x + base + (int64_t)r * C, v); gn_load8(dy + base + (int64_t)r * C, d);
        if (RELU && YMASK) o = mask[((int64_t)n * HW + r) * cbn + cb];
        GN_RED_ROW(v, d, o)
    }
#undef GN_RED_ROW
    float* mine = lpart + ((size_t)rsub * C + cb * 8) * 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) { mine[2 * j] = sa[j]; mine[2 * j + 1] = sb[j]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        float t = 0.f;
        for (int k = 0; k < rstep; ++k) t += lpart[(size_t)k * 2 * C + i];
        atomicAdd(ab + (int64_t)n * C * 2 + i, t);
        if (!defer_affine) atomicAdd(((i & 1) ? dgamma : dbeta) + (i >> 1), t);
    }
}

// dgamma[c] += sum_n ab[n][c][1], dbeta[c] += sum_n ab[n][c][0]: the affine gradients are the frame sums of the per-frame partials the
// reduction pass leaves in `ab` anyway.  Used (default; MAED_GN_DEFER_AFFINE=0 switches it off) instead of the 2C atomics every workgroup of the reduction
// pass otherwise sends to the same 2C addresses (~768 workgroups per layer: on the 14x14 and 28x28 layers the serialised atomics,
// not HBM, set that pass's ~23 us floor -- profiles/r01_rocprofv3_last_step_kernel_sequence_v9.txt).
__global__ __launch_bounds__(256) void gn_affine_grad_kernel(const float* __restrict__ ab, float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int C) {
    colsum_add(ab, N, 2 * C, [=](int i) { return ((i & 1) ? dgamma : dbeta) + (i >> 1); });      // ab[n][c][0|1] = (dbeta, dgamma) partials
}

// The same closing sum for MANY layers in one launch (maed_gn_affine_grad_batch): workgroup b -> (layer, block of 64 columns of its (N, 2C) partial matrix) through
// a prefix table that travels in the kernel arguments with the layers' pointers; 64 columns x 4 row lanes per workgroup over all N rows, one LDS fold, one atomic
// per column and layer.
#define GN_BATCH_MAX 64
struct GnAffineBatch {
    maed_gn_affine_item it[GN_BATCH_MAX];
    int first[GN_BATCH_MAX + 1];           // first workgroup of layer i; first[count] = grid size
    int count;
};
__global__ __launch_bounds__(256) void gn_affine_grad_batch_kernel(GnAffineBatch t) {
    __shared__ float fold[4][64];
    int i = 0;
    while (i + 1 < t.count && (int)blockIdx.x >= t.first[i + 1]) ++i;
    const maed_gn_affine_item it = t.it[i];
    const int cols = 2 * it.C, c = ((int)blockIdx.x - t.first[i]) * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    float s = 0.f;
    if (c < cols)
        for (int r = rl; r < it.N; r += 4) s += it.ab[(int64_t)r * cols + c];
    fold[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < cols) {
        // (an atomic: two layers of one launch may share their parameters -- a norm module applied twice in a pass)
        atomicAdd(((c & 1) ? it.dgamma : it.dbeta) + (c >> 1), (fold[0][threadIdx.x] + fold[1][threadIdx.x]) + (fold[2][threadIdx.x] + fold[3][threadIdx.x]));
    }
}
extern "C" int maed_gn_affine_grad_batch(const maed_gn_affine_item* items, int count, void* stream) {
    MAED_CHECK_ARG(count >= 0 && (items || count == 0), MAED_ERR_ARG, "gn_affine_grad_batch: null table");
    for (int base = 0; base < count; base += GN_BATCH_MAX) {
        GnAffineBatch t;
        t.count = count - base < GN_BATCH_MAX ? count - base : GN_BATCH_MAX;
        int wg = 0;
        for (int i = 0; i < t.count; ++i) {
            const maed_gn_affine_item& it = items[base + i];
            MAED_CHECK_ARG(it.ab && it.dgamma && it.dbeta && it.N > 0 && it.C > 0, MAED_ERR_ARG, "gn_affine_grad_batch: item %d: null pointer or empty extent", base + i);
            t.it[i] = it; t.first[i] = wg;
            wg += (2 * it.C + 63) / 64;
        }
        t.first[t.count] = wg;
        hipLaunchKernelGGL(gn_affine_grad_batch_kernel, dim3((unsigned)wg), dim3(256), 0, (hipStream_t)stream, t);
    }
    MAED_CHECK_LAUNCH("gn_affine_grad_batch");
    return MAED_OK;
}

// ---- backward pass 2: dx = rstd * (gamma*dy_eff - m1 - xhat*m2); optional d_res = dy_eff ------------------------
template <typename T, bool RES, bool RELU>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ x, const uint8_t* __restrict__ mask, const T* __restrict__ dy,
                                                           const double* __restrict__ sums, const float* __restrict__ ab,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           T* __restrict__ dx, T* __restrict__ dres,
                                                           int HW, int C, float eps, int rows_per_wg) {
    __shared__ float lmu[GN_G], lrs[GN_G], lm1[GN_G], lm2[GN_G];
    const int n = blockIdx.y, cpg = C / GN_G;
    if (threadIdx.x < GN_G) {
        const double cnt = (double)HW * cpg;
        const double m = sums[((int64_t)n * GN_G + threadIdx.x) * 2] / cnt;
        double var = sums[((int64_t)n * GN_G + threadIdx.x) * 2 + 1] / cnt - m * m;
        if (var < 0.0) var = 0.0;
        lmu[threadIdx.x] = (float)m; lrs[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
        float m1 = 0.f, m2 = 0.f;
        for (int c = threadIdx.x * cpg; c < (threadIdx.x + 1) * cpg; ++c) {
            m1 = fmaf(gamma[c], ab[((int64_t)n * C + c) * 2], m1);
            m2 = fmaf(gamma[c], ab[((int64_t)n * C + c) * 2 + 1], m2);
        }
        lm1[threadIdx.x] = m1 / (float)cnt; lm2[threadIdx.x] = m2 / (float)cnt;
    }
    __syncthreads();
    const int cbn = C / 8, cb = threadIdx.x % cbn, rsub = threadIdx.x / cbn, rstep = 256 / cbn;
    float mu[8], rs[8], gg[8], be[8], m1[8], m2[8];
    ld8(gamma + cb * 8, gg); ld8(beta + cb * 8, be);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int g = (cb * 8 + j) / cpg; mu[j] = lmu[g]; rs[j] = lrs[g]; m1[j] = lm1[g]; m2[j] = lm2[g]; }
    const int r0 = blockIdx.x * rows_per_wg, r1 = min(HW, r0 + rows_per_wg);
    const int64_t base = ((int64_t)n * HW) * C + cb * 8;
    for (int r = r0 + rsub; r < r1; r += rstep) {
        float v[8], d[8], o[8];
        gn_load8(x + base + (int64_t)r * C, v); gn_load8(dy + base + (int64_t)r * C, d);
        if (RELU) {
            if (RES) { const uint32_t bits = mask[((int64_t)n * HW + r) * cbn + cb];   // out = GN(x) + residual: the forward's bit mask
#pragma unroll
                for (int j = 0; j < 8; ++j) d[j] = ((bits >> j) & 1u) ? d[j] : 0.f;
            } else {                                                             // out = GN(x): recompute the mask, one tensor less to read
#pragma unroll
                for (int j = 0; j < 8; ++j) d[j] = fmaf((v[j] - mu[j]) * rs[j], gg[j], be[j]) > 0.f ? d[j] : 0.f;
            }
        }
        if (RES && dres) st8(dres + base + (int64_t)r * C, d);          // (dres == nullptr: the consumer applies the bit mask to dy itself, MAED_EPI_ADD)
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rs[j] * (gg[j] * d[j] - m1[j] - (v[j] - mu[j]) * rs[j] * m2[j]);
        st8(dx + base + (int64_t)r * C, o);
    }
}

// ---- backward in ONE pass over x and dy (round 4) -------------------------------------------------------------------------------------------------
// The two kernels above read x and dy twice: five tensor streams (2 + 2 reads, 1 write) for a result that needs three.  Nothing of a frame's dx can be written
// before the frame's 2 x 32 group sums are complete -- but the operands do not have to go back to memory in between: here the S workgroups that share a
// frame each load their slice of x and dy ONCE into registers (CH chunks of 8 channels per thread and tensor: 64 VGPRs of payload, two 512-thread workgroups
// per CU hold 2 x 128 KB), reduce it, add their per-channel partials to ab[n] with device-scope atomics, meet at a per-frame arrival counter, read the frame's
// totals back (returning atomics: served where the adds were performed, no L2 of another XCD in between) and apply straight from the registers.
// HBM traffic 5 -> 3 streams.  The workgroups of a frame are consecutive in dispatch order (blockIdx.x = slice), a workgroup waits only for its own frame's
// slices, and a frame is at most 64 workgroups against 512 resident ones: every waiting workgroup's peers are either resident or next in the queue.
// `sync`: one arrival counter per frame, ZERO at launch (the caller hands in a slice of the same per-pass zero-filled arena as `ab`).
// phase: 0 = the whole thing (GPU); 1 = reduction half only / 2 = apply half only, totals from ab -- the host simulator runs workgroups one after another, so
// its launcher issues the two halves as two launches (tests/hostsim: everything but the spin itself is then checked on the host).
#define GN1_NT 512
template <typename T> struct GnChunk;
template <> struct GnChunk<bf16> { uint4 v; };
template <> struct GnChunk<float> { float4 a, b; };
__device__ __forceinline__ void gn_chunk_ld(const bf16* p, GnChunk<bf16>& c) { c.v = *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void gn_chunk_ld(const float* p, GnChunk<float>& c) { c.a = *reinterpret_cast<const float4*>(p); c.b = *reinterpret_cast<const float4*>(p + 4); }
__device__ __forceinline__ void gn_chunk_st(bf16* p, const GnChunk<bf16>& c) { *reinterpret_cast<uint4*>(p) = c.v; }
__device__ __forceinline__ void gn_chunk_st(float* p, const GnChunk<float>& c) { *reinterpret_cast<float4*>(p) = c.a; *reinterpret_cast<float4*>(p + 4) = c.b; }
__device__ __forceinline__ void gn_chunk_zero(GnChunk<bf16>& c) { c.v = make_uint4(0u, 0u, 0u, 0u); }
__device__ __forceinline__ void gn_chunk_zero(GnChunk<float>& c) { c.a = make_float4(0.f, 0.f, 0.f, 0.f); c.b = c.a; }
__device__ __forceinline__ void gn_chunk_get(const GnChunk<bf16>& c, float (&o)[8]) {
    o[0] = __uint_as_float(c.v.x << 16); o[1] = __uint_as_float(c.v.x & 0xffff0000u); o[2] = __uint_as_float(c.v.y << 16); o[3] = __uint_as_float(c.v.y & 0xffff0000u);
    o[4] = __uint_as_float(c.v.z << 16); o[5] = __uint_as_float(c.v.z & 0xffff0000u); o[6] = __uint_as_float(c.v.w << 16); o[7] = __uint_as_float(c.v.w & 0xffff0000u);
}
__device__ __forceinline__ void gn_chunk_get(const GnChunk<float>& c, float (&o)[8]) {
    o[0] = c.a.x; o[1] = c.a.y; o[2] = c.a.z; o[3] = c.a.w; o[4] = c.b.x; o[5] = c.b.y; o[6] = c.b.z; o[7] = c.b.w;
}
__device__ __forceinline__ void gn_chunk_put(GnChunk<bf16>& c, const float (&o)[8]) {
    c.v = make_uint4(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7]));
}
__device__ __forceinline__ void gn_chunk_put(GnChunk<float>& c, const float (&o)[8]) {
    c.a = make_float4(o[0], o[1], o[2], o[3]); c.b = make_float4(o[4], o[5], o[6], o[7]);
}
// GPC: groups an 8-channel chunk spans (C / 32 channels per group: 1 for C >= 256, 2 for C = 128, 4 for C = 64); CH: chunks a thread keeps per tensor.
// What crosses the frame barrier is only what the apply step needs: 2 x 32 gamma-weighted GROUP sums per workgroup (64 atomics out, 64 returning atomics back),
// in sync[n][16..79]; the per-channel partials ab[n] (dgamma / dbeta: read by the closing column-sum kernel, not here) are added while lane 0 waits at the counter.
#define GN1_SYNC_WORDS 80                     /* per frame: [0] arrival counter, [16..79] group sums (their own cache lines) */
template <typename T, bool RELU, bool YMASK, int GPC, int CH, int NT>
__global__ __launch_bounds__(NT, 4) void gn_bwd_onepass_kernel(const T* __restrict__ x, const uint8_t* __restrict__ mask, const T* __restrict__ dy,
                                                                   const double* __restrict__ sums, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float* ab /* [N][C][2], zero at launch */, T* __restrict__ dx, T* __restrict__ dres,
                                                                   uint32_t* sync /* [N][GN1_SYNC_WORDS], zero at launch */, int HW, int C, float eps, int S,
                                                                   int rows_per_wg, int phase, int skew, uint32_t* fault /* maed_fault_word() */) {
    MAED_DYN_SHARED(float, lpart);       // [NT / cbn][C][2] per-row-lane partials (32 KB)
    __shared__ float lmu[GN_G], lrs[GN_G], lgrp[GN_G * 2], lb[GN_G], lk[GN_G];
    __shared__ int lfail;
    const int n = blockIdx.y, tid = threadIdx.x, cpg = C / GN_G;
    if (tid < GN_G) {
        const double cnt = (double)HW * cpg;
        const double m = sums[((int64_t)n * GN_G + tid) * 2] / cnt;
        double var = sums[((int64_t)n * GN_G + tid) * 2 + 1] / cnt - m * m;
        if (var < 0.0) var = 0.0;
        lmu[tid] = (float)m; lrs[tid] = (float)(1.0 / sqrt(var + (double)eps));
    }
    if (tid < GN_G * 2) lgrp[tid] = 0.f;
    if (tid == 0) lfail = 0;
    __syncthreads();
    const int cbn = C / 8, cb = tid % cbn, rsub = tid / cbn, rstep = NT / cbn;
    constexpr int JS = 8 / GPC;                                    // channels of a chunk that share a group
    const int g0 = cb * 8 / cpg;
    float mu[GPC], rs[GPC];
#pragma unroll
    for (int q = 0; q < GPC; ++q) { mu[q] = lmu[g0 + q]; rs[q] = lrs[g0 + q]; }
    const int r0 = blockIdx.x * rows_per_wg, r1 = min(HW, r0 + rows_per_wg);
    const int64_t base = ((int64_t)n * HW) * C + cb * 8;
    float* const gsum = (float*)(sync + (int64_t)n * GN1_SYNC_WORDS + 16);
#ifndef MAED_HOSTSIM
    // phase skew (multi-round launches only): the workgroups of a frame move in lockstep by construction, and every frame of the first residency round starts at
    // the same instant -- all of them stream, then all of them sit in their barrier, round after round (measured: the barrier's ~5 us stay fully exposed per
    // round).  Odd frames of the FIRST round start `skew` sleep units late: from then on odd and even frames alternate between streaming and waiting.
    if (skew > 0 && (n & 1) && (int)(blockIdx.y * gridDim.x + blockIdx.x) < 512)
        for (int i = 0; i < skew; ++i) __builtin_amdgcn_s_sleep(100);
#endif

    // ---- all loads of the slice in flight at once ----
    GnChunk<T> xr[CH], dr[CH];
    uint32_t mb[(CH + 3) / 4];
#pragma unroll
    for (int k = 0; k < (CH + 3) / 4; ++k) mb[k] = 0u;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int r = r0 + rsub + k * rstep;
        if (r < r1) {
            gn_chunk_ld(x + base + (int64_t)r * C, xr[k]);
            gn_chunk_ld(dy + base + (int64_t)r * C, dr[k]);
            if (RELU && YMASK) mb[k >> 2] |= (uint32_t)mask[((int64_t)n * HW + r) * cbn + cb] << (8 * (k & 3));
        } else { gn_chunk_zero(xr[k]); gn_chunk_zero(dr[k]); }
    }
    // ---- reduction over the slice; dr becomes the MASKED dy (exact: an element or zero) ----
    {
        float sa[8], sb[8], ga[8], be[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { sa[j] = 0.f; sb[j] = 0.f; ga[j] = 0.f; be[j] = 0.f; }
        if (RELU && !YMASK) { ld8(gamma + cb * 8, ga); ld8(beta + cb * 8, be); }     // (otherwise gamma is needed only behind the loop: 8 registers less across it)
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (r0 + rsub + k * rstep >= r1) continue;
            float v[8], d[8];
            gn_chunk_get(xr[k], v); gn_chunk_get(dr[k], d);
            const uint32_t bits = (mb[k >> 2] >> (8 * (k & 3))) & 0xffu;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = (v[j] - mu[j / JS]) * rs[j / JS];
                if (RELU) d[j] = (YMASK ? ((bits >> j) & 1u) != 0u : fmaf(xh, ga[j], be[j]) > 0.f) ? d[j] : 0.f;
                sa[j] += d[j]; sb[j] = fmaf(d[j], xh, sb[j]);
            }
            if (RELU) gn_chunk_put(dr[k], d);
        }
        if (!(RELU && !YMASK)) ld8(gamma + cb * 8, ga);
        if (phase != 2) {
            float* mine = lpart + ((size_t)rsub * C + cb * 8) * 2;
#pragma unroll
            for (int j = 0; j < 8; ++j) { mine[2 * j] = sa[j]; mine[2 * j + 1] = sb[j]; }
            // gamma-weighted group partials of this thread -> the workgroup's 2 x 32 table
#pragma unroll
            for (int q = 0; q < GPC; ++q) {
                float p1 = 0.f, p2 = 0.f;
#pragma unroll
                for (int j = q * JS; j < (q + 1) * JS; ++j) { p1 = fmaf(ga[j], sa[j], p1); p2 = fmaf(ga[j], sb[j], p2); }
                // narrow layers (C < 512): 64 / cbn lanes of a wave hold the same chunk column -- fold them first (C = 64: 64 threads of the workgroup would
                // otherwise queue at each of 8 x 8 LDS addresses: measured slower than the two-pass kernels)
                for (int m = cbn; m < 64; m <<= 1) { p1 += __shfl_xor(p1, m, 64); p2 += __shfl_xor(p2, m, 64); }
                if ((tid & 63) < cbn) { atomicAdd(&lgrp[(g0 + q) * 2], p1); atomicAdd(&lgrp[(g0 + q) * 2 + 1], p2); }
            }
        }
    }
    __syncthreads();
    if (phase != 2) {
        if (tid < GN_G * 2 && (S > 1 || phase == 1)) atomicAdd(gsum + tid, lgrp[tid]);
        if (phase == 0 && S > 1) {
            MAED_WAIT_VMCNT0();                                    // the 64 group-sum atomics have been performed
            __syncthreads();
            if (tid == 0 && !maed_frame_arrive_and_wait(sync + (int64_t)n * GN1_SYNC_WORDS, (uint32_t)S)) { lfail = 1; maed_report_fault(fault); }
        }
        // per-channel partials for dgamma / dbeta (nobody in this kernel reads them): under lane 0's wait
        for (int i = tid; i < 2 * C; i += NT) {
            float t = 0.f;
            for (int k = 0; k < rstep; ++k) t += lpart[(size_t)k * 2 * C + i];
            atomicAdd(ab + (int64_t)n * C * 2 + i, t);
        }
        if (phase == 1) return;
        __syncthreads();
    }
    if (tid < GN_G * 2 && (S > 1 || phase == 2)) lgrp[tid] = maed_coherent_read(gsum + tid);
    __syncthreads();
    if (tid < GN_G) {
        const float cnt = (float)((double)HW * cpg);
        const float poison = lfail ? __uint_as_float(0x7fc00000u) : 0.f;
        lb[tid] = lrs[tid] * (lgrp[2 * tid] / cnt) + poison; lk[tid] = lrs[tid] * (lgrp[2 * tid + 1] / cnt);
    }
    __syncthreads();
    // ---- apply from the registers: dx = rstd * (gamma * dy_eff - m1 - xhat * m2) ----
    float A[8], B[GPC], K[GPC];
    {
        float gg[8];
        ld8(gamma + cb * 8, gg);
#pragma unroll
        for (int j = 0; j < 8; ++j) A[j] = rs[j / JS] * gg[j];
#pragma unroll
        for (int q = 0; q < GPC; ++q) { B[q] = lb[g0 + q]; K[q] = lk[g0 + q]; }
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int r = r0 + rsub + k * rstep;
        if (r >= r1) continue;
        float v[8], d[8], o[8];
        gn_chunk_get(xr[k], v); gn_chunk_get(dr[k], d);
        if (dres) gn_chunk_st(dres + base + (int64_t)r * C, dr[k]);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaf(A[j], d[j], -fmaf((v[j] - mu[j / JS]) * rs[j / JS], K[j / JS], B[j / JS]));
        GnChunk<T> oc;
        gn_chunk_put(oc, o);
        gn_chunk_st(dx + base + (int64_t)r * C, oc);
    }
}

static int gn_check(int C, int HW, const char* who) {
    MAED_CHECK_ARG(C % GN_G == 0 && C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0, MAED_ERR_SHAPE, "%s: C=%d unsupported (need C%%32==0, C/8 a power of two <= 256)", who, C);
    MAED_CHECK_ARG(HW > 0, MAED_ERR_SHAPE, "%s: HW=%d", who, HW);
    return MAED_OK;
}
static int gn_rows_per_wg(int N, int HW, int C, int target_wgs = 2048) {
    // aim for ~target_wgs workgroups overall; at least one pass of the row tile (256 / (C/8) rows)
    const int tile = 256 / (C / 8);
    int chunks = (target_wgs + N - 1) / N;
    int rows = (HW + chunks - 1) / chunks;
    rows = (rows + tile - 1) / tile * tile;
    return rows < tile ? tile : rows;
}

static int groupnorm_fwd(const void* x, const void* residual, const float* gamma, const float* beta, void* y, double* sums,
                         uint8_t* relu_mask, int N, int HW, int C, float eps, int relu, int dtype, int sums_zeroed, void* twin_x, void* twin_y, void* stream) {
    MAED_CHECK_ARG(x && gamma && beta && y && sums, MAED_ERR_ARG, "groupnorm_fwd: null pointer");
    MAED_PROPAGATE(gn_check(C, HW, "groupnorm_fwd"));
    if (N <= 0) return MAED_OK;
    hipStream_t s = (hipStream_t)stream;
    const int rows = gn_rows_per_wg(N, HW, C);
    dim3 grid((HW + rows - 1) / rows, N);
    if (!sums_zeroed) MAED_HIP(hipMemsetAsync(sums, 0, (size_t)N * GN_G * 2 * sizeof(double), s), "groupnorm_fwd: memset");
    MAED_DISPATCH_DTYPE(dtype, T, {
        if (sums_zeroed != 2)       // 2: the producing convolution's epilogue accumulated the statistics already (maed_conv1x1_fwd / maed_conv3x3_fwd)
            hipLaunchKernelGGL((gn_stats_kernel<T>), grid, dim3(256), 0, s, (const T*)x, sums, HW, C, rows);
        if (residual && relu) hipLaunchKernelGGL((gn_apply_kernel<T, true, true>), grid, dim3(256), 0, s, (const T*)x, (const T*)residual, sums, gamma, beta, (T*)y, relu_mask, HW, C, eps, rows, (bf16*)twin_x, (bf16*)twin_y);
        else if (residual) hipLaunchKernelGGL((gn_apply_kernel<T, true, false>), grid, dim3(256), 0, s, (const T*)x, (const T*)residual, sums, gamma, beta, (T*)y, relu_mask, HW, C, eps, rows, (bf16*)twin_x, (bf16*)twin_y);
        else if (relu) hipLaunchKernelGGL((gn_apply_kernel<T, false, true>), grid, dim3(256), 0, s, (const T*)x, (const T*)nullptr, sums, gamma, beta, (T*)y, relu_mask, HW, C, eps, rows, (bf16*)twin_x, (bf16*)twin_y);
        else hipLaunchKernelGGL((gn_apply_kernel<T, false, false>), grid, dim3(256), 0, s, (const T*)x, (const T*)nullptr, sums, gamma, beta, (T*)y, relu_mask, HW, C, eps, rows, (bf16*)twin_x, (bf16*)twin_y);
    });
    MAED_CHECK_LAUNCH("groupnorm_fwd");
    return MAED_OK;
}

extern "C" int maed_groupnorm_fwd(const void* x, const void* residual, const float* gamma, const float* beta, void* y, double* sums,
                                  uint8_t* relu_mask, int N, int HW, int C, float eps, int relu, int dtype, int sums_zeroed, void* stream) {
    return groupnorm_fwd(x, residual, gamma, beta, y, sums, relu_mask, N, HW, C, eps, relu, dtype, sums_zeroed, nullptr, nullptr, stream);
}
// the fp32 forward that also leaves bf16 twins of its input and its result ("bf16x3 forward / bf16 backward from bf16 twins": the backward reads x, the next
// layer's backward reads y -- both as bf16): two more 2-byte stores per element from registers the pass holds anyway, instead of a cast pass per tensor
extern "C" int maed_groupnorm_fwd_twin(const void* x, const void* residual, const float* gamma, const float* beta, void* y, double* sums,
                                       uint8_t* relu_mask, int N, int HW, int C, float eps, int relu, int sums_zeroed, void* twin_x, void* twin_y, void* stream) {
    MAED_CHECK_ARG(is_aligned(twin_x, 16) && is_aligned(twin_y, 16), MAED_ERR_ALIGN, "groupnorm_fwd_twin: 16-B alignment");
    return groupnorm_fwd(x, residual, gamma, beta, y, sums, relu_mask, N, HW, C, eps, relu, MAED_F32, sums_zeroed, twin_x, twin_y, stream);
}

extern "C" int maed_groupnorm_bwd(const void* x, const uint8_t* relu_mask, const void* dy, const double* sums, const float* gamma, const float* beta,
                                  void* dx, void* dres, float* dgamma, float* dbeta, float* ab_scratch, int N, int HW, int C, float eps,
                                  int relu, int dtype, int ab_zeroed, uint32_t* frame_sync, void* aux_stream, void* stream) {
    MAED_CHECK_ARG(x && dy && sums && gamma && beta && dx && ab_scratch, MAED_ERR_ARG, "groupnorm_bwd: null pointer");
    MAED_CHECK_ARG((dgamma != nullptr) == (dbeta != nullptr), MAED_ERR_ARG, "groupnorm_bwd: dgamma and dbeta go together (both NULL: the caller folds ab_scratch with maed_gn_affine_grad_batch)");
    MAED_CHECK_ARG(!(relu && dres) || relu_mask, MAED_ERR_ARG, "groupnorm_bwd: the forward's relu_mask is needed when a residual was added before the ReLU");
    MAED_PROPAGATE(gn_check(C, HW, "groupnorm_bwd"));
    if (N <= 0) return MAED_OK;
    hipStream_t s = (hipStream_t)stream;
    const int rows = gn_rows_per_wg(N, HW, C);
    dim3 grid((HW + rows - 1) / rows, N);
    // the reduction pass ends with 4C atomics per workgroup: fewer, fatter workgroups (~768: 3 per CU) keep it HBM-bound
    const int rrows = gn_rows_per_wg(N, HW, C, 768);
    dim3 rgrid((HW + rrows - 1) / rrows, N);
    if (!ab_zeroed) MAED_HIP(hipMemsetAsync(ab_scratch, 0, (size_t)N * C * 2 * sizeof(float), s), "groupnorm_bwd: memset");
    const size_t lds = (size_t)(256 / (C / 8)) * 2 * C * sizeof(float);
    // dgamma / dbeta from per-workgroup partials + a closing column sum instead of contended atomics in the reduction pass
    // (timed on MI355X, profiles/r02_call2_steady_*.csv: reduction passes 2.15 -> 1.42 ms per step + 0.31 ms closing kernels)
    constexpr int defer = 1;
    // one pass over x and dy (gn_bwd_onepass_kernel) when the caller provides the per-frame arrival counters (zero at launch) and a frame fits 64 workgroups
    // chunks per thread and tensor: 64 payload VGPRs (bf16: 8 chunks, f32: 4); one fewer where the ReLU mask is recomputed from x (gamma and beta live through
    // the reduction: 8 chunks spill 7-14 VGPRs to scratch at the 128-VGPR budget of two 512-thread workgroups per CU)
    const bool ymask = relu && relu_mask;      // residual added before the ReLU: the forward's bit mask (dres may be NULL: not materialised)
    const int cpg = C / GN_G, cbn1 = C / 8, ch1 = dtype == MAED_BF16 ? ((relu && !ymask) ? 7 : 8) : 4;
    int S1 = 0, rows1 = 0;
    const int nt1 = maed_opt(MAED_OPT_GN_BWD_ONEPASS) == 2 ? 256 : GN1_NT;      // (2: experiment -- 256-thread workgroups, four per CU)
    // a frame barrier that timed out earlier in this process (shared GPU, preemption: the peers of a frame were not co-resident) poisoned that call's result with NaN
    // and raised the fault word: from then on the two-pass kernels run (maed_device_faults() / maed_last_error() tell the host)
    uint32_t* const fault1 = maed_fault_word();
    bool onepass = frame_sync && maed_opt(MAED_OPT_GN_BWD_ONEPASS) && cpg >= 2 && cbn1 <= nt1 && !maed_fault_seen("groupnorm_bwd");
    if (onepass) {
        const int rstep = nt1 / cbn1, cap = ch1 * rstep;
        S1 = (HW + cap - 1) / cap;
        rows1 = (((HW + S1 - 1) / S1) + rstep - 1) / rstep * rstep;      // balanced slices, whole row steps
        onepass = S1 <= 64;
    }
#define GN_RED(RELU_, YM_) hipLaunchKernelGGL((gn_bwd_reduce_kernel<T, RELU_, YM_>), rgrid, dim3(256), lds, s, (const T*)x, relu_mask, (const T*)dy, \
        sums, gamma, beta, ab_scratch, dgamma, dbeta, HW, C, eps, rrows, defer)
#define GN_APP(RES_, RELU_) hipLaunchKernelGGL((gn_bwd_apply_kernel<T, RES_, RELU_>), grid, dim3(256), 0, s, (const T*)x, relu_mask, (const T*)dy, \
        sums, ab_scratch, gamma, beta, (T*)dx, (T*)dres, HW, C, eps, rows)
#define GN_ONE4(RELU_, YM_, GPC_, CH_, PH_, NT_) hipLaunchKernelGGL((gn_bwd_onepass_kernel<T, RELU_, YM_, GPC_, CH_, NT_>), dim3(S1, N), dim3(NT_), lds1, s, (const T*)x, relu_mask, \
        (const T*)dy, sums, gamma, beta, ab_scratch, (T*)dx, (T*)dres, frame_sync, HW, C, eps, (maed_opt(MAED_OPT_GN_BWD_ONEPASS) == 3 ? 1 : S1), rows1, PH_, skew1, fault1)
#define GN_ONE3(RELU_, YM_, GPC_, CH_, PH_) do { if (nt1 == 256) GN_ONE4(RELU_, YM_, GPC_, CH_, PH_, 256); else GN_ONE4(RELU_, YM_, GPC_, CH_, PH_, GN1_NT); } while (0)
#define GN_ONE2(RELU_, YM_, CH_, PH_) do { if (cpg >= 8) GN_ONE3(RELU_, YM_, 1, CH_, PH_); else if (cpg == 4) GN_ONE3(RELU_, YM_, 2, CH_, PH_); else GN_ONE3(RELU_, YM_, 4, CH_, PH_); } while (0)
#define GN_ONE(PH_) do { constexpr int CH_ = sizeof(T) == 2 ? 8 : 4, CHM_ = sizeof(T) == 2 ? 7 : 4; \
        if (!relu) GN_ONE2(false, false, CH_, PH_); else if (ymask) GN_ONE2(true, true, CH_, PH_); else GN_ONE2(true, false, CHM_, PH_); } while (0)
    const size_t lds1 = (size_t)(nt1 / cbn1) * 2 * C * sizeof(float);
    // phase skew of the odd frames of the first residency round (see the kernel), launches of three or more rounds: 2 x 100 sleep units ~ 5 us measured best
    // (profiles/r04_gn_bwd_onepass_micro.txt: 56x56x256 171 -> 158 us, 28x28x512 98 -> 86; 4 units the same, 6 worse than none); MAED_OPT_GN_BWD_ONEPASS >= 10
    // overrides with (value - 10) units (sweep knob)
    const int opt1 = maed_opt(MAED_OPT_GN_BWD_ONEPASS);
    const int skew1 = (int64_t)S1 * N >= 1536 ? (opt1 >= 10 ? opt1 - 10 : 2) : 0;
    MAED_DISPATCH_DTYPE(dtype, T, {
        if (onepass) {
#ifdef MAED_HOSTSIM
            GN_ONE(1); GN_ONE(2);       // workgroups run one after another on the host: the two halves as two launches
#else
            GN_ONE(0);
#endif
        } else if (!relu) GN_RED(false, false); else if (ymask) GN_RED(true, true); else GN_RED(true, false);
        if (defer && dgamma) {
            // dgamma / dbeta are read by nobody before the caller joins aux_stream (ops.side_stream_join): the small column-sum kernel leaves the
            // dy -> dx chain and runs beside the apply pass
            hipStream_t sa = s;
            if (aux_stream) {      // (the library's one fence ring: maed_init_runtime, block.hip)
                MAED_PROPAGATE(maed_stream_fence(s, aux_stream));
                sa = (hipStream_t)aux_stream;
            }
            hipLaunchKernelGGL(gn_affine_grad_kernel, dim3((2 * C + 63) / 64, (N + 63) / 64), dim3(256), 0, sa, ab_scratch, dgamma, dbeta, N, C);
        }
        if (onepass) { }
        else if (ymask) GN_APP(true, true); else if (dres) GN_APP(true, false); else if (relu) GN_APP(false, true); else GN_APP(false, false);
    });
#undef GN_RED
#undef GN_APP
#undef GN_ONE
#undef GN_ONE2
#undef GN_ONE3
#undef GN_ONE4
    MAED_CHECK_LAUNCH("groupnorm_bwd");
    return MAED_OK;
}

// ---------------------------------------------------------------------------------------------------------
// MaxPool2dSame(3, stride 2) of the stem (resnetv2.py:61-72: TF 'SAME' padding with -inf, then max_pool2d) on channels_last.
// One pass forward (no padded copy) that also records the winning tap; backward GATHERS (each input pixel looks at the <= 4
// windows that cover it) instead of ATen's scatter -- 341 -> ~100 us at cfg3.  Ties / NaN follow ATen: the first strictly
// greater value in (kh, kw) scan order wins, a NaN always wins.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool3s2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx, int N, int H, int W,
                                                             int C, int Ho, int Wo, int top, int left) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one thread: 8 channels of one output pixel
    const int cb = C / 8;
    if (i >= (int64_t)N * Ho * Wo * cb) return;
    const int c8 = (int)(i % cb) * 8;
    const int64_t pix = i / cb;
    const int wo = (int)(pix % Wo), ho = (int)((pix / Wo) % Ho), n = (int)(pix / ((int64_t)Wo * Ho));
    float m[8]; int am[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { m[j] = -INFINITY; am[j] = 0; }
    bool first = true;
    for (int kh = 0; kh < 3; ++kh) {
        const int h = 2 * ho - top + kh;
        if (h < 0 || h >= H) continue;
        for (int kw = 0; kw < 3; ++kw) {
            const int w = 2 * wo - left + kw;
            if (w < 0 || w >= W) continue;
            float v[8];
            ld8(x + (((int64_t)n * H + h) * W + w) * C + c8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (first || v[j] > m[j] || v[j] != v[j]) { m[j] = v[j]; am[j] = kh * 3 + kw; }
            first = false;
        }
    }
    st8(y + i * 8, m);
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { lo |= (uint32_t)am[j] << (8 * j); hi |= (uint32_t)am[4 + j] << (8 * j); }
    *reinterpret_cast<uint2*>(idx + i * 8) = make_uint2(lo, hi);
}

// Backward: a thread owns 8 channels of a 2 x 2 block of INPUT pixels chosen so that its four pixels lie under the same (at most four) windows -- rows
// 2k - top, 2k - top + 1 see windows k - 1 and k -- and reads every window's gradient and winning taps ONCE for the four of them (round 6: one thread per input
// pixel read them four times over: 99 us for 282 MB of compulsory traffic).  Per pixel the windows are added in the same order as before: the same bits.
template <typename T>
__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx, T* __restrict__ dx, int N, int H,
                                                             int W, int C, int Ho, int Wo, int top, int left) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int cb = C / 8, Hb = (H + top + 1) >> 1, Wb = (W + left + 1) >> 1;
    if (i >= (int64_t)N * Hb * Wb * cb) return;
    const int c8 = (int)(i % cb) * 8;
    const int64_t blk = i / cb;
    const int l = (int)(blk % Wb), k = (int)((blk / Wb) % Hb), n = (int)(blk / ((int64_t)Wb * Hb));
    float g[2][2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 8; ++j) g[a][b][j] = 0.f;
#pragma unroll
    for (int wh = 0; wh < 2; ++wh) {                       // window rows k - 1, k (ascending, as the per-pixel loop of rounds 2-5)
        const int ho = k - 1 + wh;
        if (ho < 0 || ho >= Ho) continue;
#pragma unroll
        for (int ww = 0; ww < 2; ++ww) {
            const int wo = l - 1 + ww;
            if (wo < 0 || wo >= Wo) continue;
            const int64_t o = ((((int64_t)n * Ho + ho) * Wo + wo) * C + c8);
            const uint2 a = *reinterpret_cast<const uint2*>(idx + o);
            float d[8];
            ld8(dy + o, d);
            // pixel (dh, dw) of the block sits at tap (2 (k - ho) + dh, 2 (l - wo) + dw) of this window: inside it for the window's own block, and for dh / dw = 0 of
            // the block after it
#pragma unroll
            for (int dh = 0; dh < 2; ++dh) {
                const int kh = 2 * (1 - wh) + dh;
                if (kh > 2) continue;
#pragma unroll
                for (int dw = 0; dw < 2; ++dw) {
                    const int kw = 2 * (1 - ww) + dw;
                    if (kw > 2) continue;
                    const uint32_t tap = (uint32_t)(kh * 3 + kw);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (((a.x >> (8 * j)) & 0xffu) == tap) g[dh][dw][j] += d[j];
                        if (((a.y >> (8 * j)) & 0xffu) == tap) g[dh][dw][4 + j] += d[4 + j];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int dh = 0; dh < 2; ++dh) {
        const int h = 2 * k - top + dh;
        if (h < 0 || h >= H) continue;
#pragma unroll
        for (int dw = 0; dw < 2; ++dw) {
            const int w = 2 * l - left + dw;
            if (w < 0 || w >= W) continue;
            st8(dx + (((int64_t)n * H + h) * W + w) * C + c8, g[dh][dw]);
        }
    }
}

static void maxpool_geom(int H, int W, int& Ho, int& Wo, int& top, int& left) {
    Ho = (H + 1) / 2; Wo = (W + 1) / 2;
    const int ph = (Ho - 1) * 2 + 3 - H, pw = (Wo - 1) * 2 + 3 - W;
    top = (ph > 0 ? ph : 0) / 2; left = (pw > 0 ? pw : 0) / 2;
}

extern "C" int maed_maxpool3s2_same_fwd(const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, int dtype, void* stream) {
    MAED_CHECK_ARG(x && y && idx, MAED_ERR_ARG, "maxpool3s2_same_fwd: null pointer");
    MAED_CHECK_ARG(N >= 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, MAED_ERR_SHAPE, "maxpool3s2_same_fwd: C=%d must be a multiple of 8", C);
    int Ho, Wo, top, left;
    maxpool_geom(H, W, Ho, Wo, top, left);
    const int64_t n = (int64_t)N * Ho * Wo * (C / 8);
    if (n == 0) return MAED_OK;
    MAED_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((maxpool3s2_fwd_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                                                      (const T*)x, (T*)y, idx, N, H, W, C, Ho, Wo, top, left));
    MAED_CHECK_LAUNCH("maxpool3s2_same_fwd");
    return MAED_OK;
}

extern "C" int maed_maxpool3s2_same_bwd(const void* dy, const uint8_t* idx, void* dx, int N, int H, int W, int C, int dtype, void* stream) {
    MAED_CHECK_ARG(dy && idx && dx, MAED_ERR_ARG, "maxpool3s2_same_bwd: null pointer");
    MAED_CHECK_ARG(N >= 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, MAED_ERR_SHAPE, "maxpool3s2_same_bwd: C=%d must be a multiple of 8", C);
    int Ho, Wo, top, left;
    maxpool_geom(H, W, Ho, Wo, top, left);
    const int64_t n = (int64_t)N * ((H + top + 1) / 2) * ((W + left + 1) / 2) * (C / 8);       // 2 x 2 blocks of input pixels
    if (n == 0) return MAED_OK;
    MAED_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((maxpool3s2_bwd_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                                                      (const T*)dy, idx, (T*)dx, N, H, W, C, Ho, Wo, top, left));
    MAED_CHECK_LAUNCH("maxpool3s2_same_bwd");
    return MAED_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Input of the 7x7 stem convolution (resnetv2.py:51-59,74-93,282-285): the clip arrives fp32 NCHW; MIOpen's solver wants the compute dtype, channels_last, and
// a SYMMETRIC padding -- TF-SAME for kernel 7 / stride 2 on 224 is 2 in front and 3 behind.  The framework did that in three passes (dtype cast, layout copy,
// zero-filled padded copy: 35 + 31 + 8 us at cfg3).  One pass here: thread per padded pixel, three plane reads (coalesced along x), one 6- or 12-byte store.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void stem_input_kernel(const float* __restrict__ x, T* __restrict__ y, int64_t n, int C, int CS, int H, int W, int Hp, int Wp, int top, int left) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one thread: all CS (<= 4) channel slots of one padded pixel (slots >= C: zero)
    if (i >= n) return;
    const int xp = (int)(i % Wp), yp = (int)((i / Wp) % Hp);
    const int64_t f = i / ((int64_t)Wp * Hp);
    const int xx = xp - left, yy = yp - top;
    const bool in = (unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H;
    if (CS == 4) {
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = (in && c < C) ? x[((f * C + c) * H + yy) * (int64_t)W + xx] : 0.f;
        st4(y + i * 4, v);
        return;
    }
    for (int c = 0; c < CS; ++c) stf(y + i * CS + c, (in && c < C) ? x[((f * C + c) * H + yy) * (int64_t)W + xx] : 0.f);
}

extern "C" int maed_stem_input(const float* x, void* y, int N, int C, int H, int W, int pad_top, int pad_bottom, int pad_left, int pad_right, int c_stride, int dtype,
                               void* stream) {
    MAED_CHECK_ARG(x && y, MAED_ERR_ARG, "stem_input: null pointer");
    MAED_CHECK_ARG(N >= 0 && C > 0 && C <= c_stride && c_stride <= 4 && H > 0 && W > 0 && pad_top >= 0 && pad_bottom >= 0 && pad_left >= 0 && pad_right >= 0, MAED_ERR_SHAPE,
                   "stem_input: bad extents (C <= c_stride <= 4)");
    MAED_CHECK_ARG(c_stride != 4 || is_aligned(y, 16), MAED_ERR_ALIGN, "stem_input: y must be 16-B aligned");
    const int Hp = H + pad_top + pad_bottom, Wp = W + pad_left + pad_right;
    const int64_t n = (int64_t)N * Hp * Wp;
    if (n == 0) return MAED_OK;
    MAED_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((stem_input_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (T*)y, n, C, c_stride, H,
                                                      W, Hp, Wp, pad_top, pad_left));
    MAED_CHECK_LAUNCH("stem_input");
    return MAED_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Pixel subsampling of a 1x1 stride-2 convolution (the downsample shortcut of stages 2 and 3, resnetv2.py:207-216: TF-SAME padding of a
// 1x1 kernel is zero for every input size, so output pixel (oy, ox) reads input pixel (2 oy, 2 ox)).  Forward packs those pixels into a
// dense (F, Ho, Wo, C) activation -- the convolution is then a plain GEMM on 1/4 of the rows, and the packed copy is also the weight
// gradient's operand; backward spreads the packed input gradient back (every other pixel is zero: one write pass, no memset).
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void subsample2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, int H, int W, int C, int Ho, int Wo) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one thread: 8 channels of one output pixel
    if (i >= n) return;
    const int cb = C / 8, c8 = (int)(i % cb) * 8;
    const int64_t pix = i / cb;
    const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho);
    const int64_t f = pix / ((int64_t)Wo * Ho);
    float v[8];
    ld8(x + ((f * H + 2 * oy) * W + 2 * ox) * C + c8, v);
    st8(y + i * 8, v);
}

template <typename T>
__global__ __launch_bounds__(256) void subsample2_bwd_kernel(const T* __restrict__ g, T* __restrict__ dx, int64_t n, int H, int W, int C, int Ho, int Wo) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one thread: 8 channels of one INPUT pixel
    if (i >= n) return;
    const int cb = C / 8, c8 = (int)(i % cb) * 8;
    const int64_t pix = i / cb;
    const int ix = (int)(pix % W), iy = (int)((pix / W) % H);
    const int64_t f = pix / ((int64_t)W * H);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!((ix | iy) & 1)) ld8(g + ((f * Ho + (iy >> 1)) * Wo + (ix >> 1)) * C + c8, v);
    st8(dx + i * 8, v);
}

extern "C" int maed_subsample2_fwd(const void* x, void* y, int F, int H, int W, int C, int dtype, void* stream) {
    MAED_CHECK_ARG(x && y, MAED_ERR_ARG, "subsample2_fwd: null pointer");
    MAED_CHECK_ARG(F >= 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, MAED_ERR_SHAPE, "subsample2_fwd: C=%d must be a multiple of 8", C);
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int64_t n = (int64_t)F * Ho * Wo * (C / 8);
    if (n == 0) return MAED_OK;
    MAED_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((subsample2_fwd_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                                                      (const T*)x, (T*)y, n, H, W, C, Ho, Wo));
    MAED_CHECK_LAUNCH("subsample2_fwd");
    return MAED_OK;
}

extern "C" int maed_subsample2_bwd(const void* g, void* dx, int F, int H, int W, int C, int dtype, void* stream) {
    MAED_CHECK_ARG(g && dx, MAED_ERR_ARG, "subsample2_bwd: null pointer");
    MAED_CHECK_ARG(F >= 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, MAED_ERR_SHAPE, "subsample2_bwd: C=%d must be a multiple of 8", C);
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int64_t n = (int64_t)F * H * W * (C / 8);
    if (n == 0) return MAED_OK;
    MAED_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((subsample2_bwd_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                                                      (const T*)g, (T*)dx, n, H, W, C, Ho, Wo));
    MAED_CHECK_LAUNCH("subsample2_bwd");
    return MAED_OK;
}
