// K1: LayerNorm forward / backward.  HBM-bound: one wave per row, 16-byte accesses, the row stays in
// registers between the statistics passes (two-pass variance like ATen, vision_transformer.py:569).
#include "common.cuh"

#define LN_MAXV 8  // float4 chunks per lane: C <= 64*4*8 = 2048
// The kernels are instantiated per chunk count (C <= 256*NV): the per-lane arrays are sized by NV, so C = 512 runs with a
// quarter of the registers of the generic NV = 8 version (backward: ~180 -> ~60 VGPRs, 2 -> 8 waves per SIMD in flight).

template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, int64_t xs,
                                                     const float* __restrict__ g, const float* __restrict__ b,
                                                     T* __restrict__ y, float* __restrict__ mean_o,
                                                     float* __restrict__ rstd_o, int64_t rows, int C, float eps, uint32_t* __restrict__ clear, int clear_words,
                                                     bf16* __restrict__ y_lo) {
    // y_lo (T = bf16 only): the output as (hi, lo) bf16 planes for the split products of the accurate mode -- y = bf16(o) is the hi plane (and the bf16 backward's twin),
    // y_lo = bf16(o - hi) (csrc/gemm_x3p.hip)
    // (piggy-backed scratch clear of the fused STE block: the arrival counters its attentive-addition kernel polls later in the same call -- a memset node of
    //  its own was a launch per block and direction on the dependent chain)
    if (clear) { const int64_t ci = (int64_t)blockIdx.x * 256 + threadIdx.x; if (ci < clear_words) clear[ci] = 0u; }
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * xs;
    float v[NV][4];
    float s = 0.f;
    const int nv = C / 4;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c4 = lane + i * 64;
        if (c4 < nv) { ld4(xr + c4 * 4, v[i]); s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c4 = lane + i * 64;
        if (c4 < nv) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    if (lane == 0) { if (mean_o) mean_o[row] = mean; if (rstd_o) rstd_o[row] = rstd; }
    T* yr = y + row * (int64_t)C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c4 = lane + i * 64;
        if (c4 < nv) {
            float gg[4], bb[4], o[4];
            ld4(g + c4 * 4, gg); ld4(b + c4 * 4, bb);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * gg[j] + bb[j];
            st4(yr + c4 * 4, o);
            if constexpr (sizeof(T) == 2) {
                if (y_lo) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] -= round_to<bf16>(o[j]);
                    st4(y_lo + row * (int64_t)C + c4 * 4, o);
                }
            }
        }
    }
}

// Each workgroup (4 waves) walks LN_ROWS_PER_WG rows; per-lane partial dgamma/dbeta stay in registers,
// are combined across the 4 waves through LDS and leave the workgroup as one atomicAdd per column.
#define LN_ROWS_PER_WG 32

// PARTIALS: instead of 2C atomics per workgroup onto the same 2C addresses (788 workgroups at cfg3: ~25k atomics per cache line, a
// serial tail of ~20 us after ~31 us of streaming), every workgroup stores its column sums to row blockIdx.x of `dg` ([nwg][2C],
// dgamma | dbeta) and ln_affine_finish_kernel adds the column totals to the gradients.  Default inside the fused block (block.hip; MAED_LN_DEFER_AFFINE=0 switches it off).
template <typename T, int NV, bool PARTIALS = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ x, int64_t xs,
                                                     const float* __restrict__ g, const float* __restrict__ mean_i,
                                                     const float* __restrict__ rstd_i, const float* __restrict__ dres,
                                                     float* __restrict__ dx, T* __restrict__ dx_twin, float* __restrict__ dg,
                                                     float* __restrict__ db, int64_t rows, int C, uint32_t* __restrict__ clear, int clear_words) {
    MAED_DYN_SHARED(float, lds);  // [2][4 waves][C]
    if (clear) { const int64_t ci = (int64_t)blockIdx.x * 256 + threadIdx.x; if (ci < clear_words) clear[ci] = 0u; }       // (piggy-backed scratch clear, as in ln_fwd_kernel)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = C / 4;
    float gg[NV][4], pg[NV][4], pb[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c4 = lane + i * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) { pg[i][j] = 0.f; pb[i][j] = 0.f; gg[i][j] = 0.f; }
        if (c4 < nv) ld4(g + c4 * 4, gg[i]);
    }
    const int64_t row0 = (int64_t)blockIdx.x * LN_ROWS_PER_WG;
    for (int r = wave; r < LN_ROWS_PER_WG; r += 4) {
        const int64_t row = row0 + r;
        if (row >= rows) break;
        const float mean = mean_i[row], rstd = rstd_i[row];
        const float* xr = x + row * xs;
        const T* dyr = dy + row * (int64_t)C;
        float xh[NV][4], gy[NV][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c4 = lane + i * 64;
            if (c4 < nv) {
                float xv[4], dv[4];
                ld4(xr + c4 * 4, xv); ld4(dyr + c4 * 4, dv);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xh[i][j] = (xv[j] - mean) * rstd;
                    gy[i][j] = dv[j] * gg[i][j];
                    s1 += gy[i][j]; s2 += gy[i][j] * xh[i][j];
                    pg[i][j] += dv[j] * xh[i][j]; pb[i][j] += dv[j];
                }
            }
        }
        s1 = wave_sum(s1) / (float)C; s2 = wave_sum(s2) / (float)C;
        float* dxr = dx + row * (int64_t)C;
        const float* drr = dres ? dres + row * (int64_t)C : nullptr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c4 = lane + i * 64;
            if (c4 < nv) {
                float o[4], rr[4] = {0.f, 0.f, 0.f, 0.f};
                if (drr) ld4(drr + c4 * 4, rr);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = rr[j] + rstd * (gy[i][j] - s1 - xh[i][j] * s2);
                st4(dxr + c4 * 4, o);
                if (dx_twin) st4(dx_twin + row * (int64_t)C + c4 * 4, o);   // compute-dtype copy: next GEMM's operand
            }
        }
    }
    float* lg = lds + (size_t)wave * C;
    float* lb = lds + (size_t)(4 + wave) * C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c4 = lane + i * 64;
        if (c4 < nv) { st4(lg + c4 * 4, pg[i]); st4(lb + c4 * 4, pb[i]); }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const float sg = (lds[c] + lds[C + c]) + (lds[2 * C + c] + lds[3 * C + c]);
        const float sb = (lds[4 * C + c] + lds[5 * C + c]) + (lds[6 * C + c] + lds[7 * C + c]);
        if constexpr (PARTIALS) {
            dg[(int64_t)blockIdx.x * 2 * C + c] = sg;
            dg[(int64_t)blockIdx.x * 2 * C + C + c] = sb;
        } else {
            atomicAdd(dg + c, sg);
            atomicAdd(db + c, sb);
        }
    }
}

__global__ __launch_bounds__(256) void ln_affine_finish_kernel(const float* __restrict__ partials, int nwg, int C, float* __restrict__ dg,
                                                               float* __restrict__ db) {
    colsum_add(partials, nwg, 2 * C, [=](int i) { return i < C ? dg + i : db + (i - C); });       // rows = [dgamma (C) | dbeta (C)]
}

// internal (block.hip): the forward that also zeroes `clear_words` 4-byte words at `clear` (clear_words <= 256 * ceil(rows / 4))
int maed_layernorm_fwd_ws(const float* x, int64_t x_row_stride, const float* gamma, const float* beta, void* y, int dtype, float* mean, float* rstd, int64_t rows,
                          int C, float eps, uint32_t* clear, int clear_words, void* stream, void* y_lo = nullptr);
extern "C" int maed_layernorm_fwd(const float* x, int64_t x_row_stride, const float* gamma, const float* beta,
                                  void* y, int dtype, float* mean, float* rstd, int64_t rows, int C, float eps,
                                  void* stream) {
    return maed_layernorm_fwd_ws(x, x_row_stride, gamma, beta, y, dtype, mean, rstd, rows, C, eps, nullptr, 0, stream);
}
int maed_layernorm_fwd_ws(const float* x, int64_t x_row_stride, const float* gamma, const float* beta, void* y, int dtype, float* mean, float* rstd, int64_t rows,
                          int C, float eps, uint32_t* clear, int clear_words, void* stream, void* y_lo) {
    MAED_CHECK_ARG(x && gamma && beta && y, MAED_ERR_ARG, "layernorm_fwd: null pointer");
    MAED_CHECK_ARG(!y_lo || (dtype == MAED_BF16 && is_aligned(y_lo, 8)), MAED_ERR_ARG, "layernorm_fwd: a lo plane goes with a bf16 (hi plane) output");
    MAED_CHECK_ARG(C > 0 && C % 4 == 0 && C <= 64 * 4 * LN_MAXV, MAED_ERR_SHAPE, "layernorm_fwd: C=%d must be a multiple of 4, <= %d", C, 64 * 4 * LN_MAXV);
    MAED_CHECK_ARG(x_row_stride % 4 == 0 && is_aligned(x, 16) && is_aligned(y, 8), MAED_ERR_ALIGN, "layernorm_fwd: x/y/stride alignment");
    if (rows == 0) return MAED_OK;
    dim3 grid((unsigned)((rows + 3) / 4));
    MAED_CHECK_ARG(!clear || (int64_t)clear_words <= (int64_t)grid.x * 256, MAED_ERR_SHAPE, "layernorm_fwd: %d words to clear exceed the grid", clear_words);
#define LN_FWD(NV_) hipLaunchKernelGGL((ln_fwd_kernel<T, NV_>), grid, dim3(256), 0, (hipStream_t)stream, x, x_row_stride, gamma, beta, (T*)y, mean, rstd, rows, C, eps, clear, clear_words, (bf16*)y_lo)
    MAED_DISPATCH_DTYPE(dtype, T, { if (C <= 512) LN_FWD(2); else if (C <= 768) LN_FWD(3); else if (C <= 1024) LN_FWD(4); else LN_FWD(8); });
#undef LN_FWD
    MAED_CHECK_LAUNCH("layernorm_fwd");
    return MAED_OK;
}

size_t maed_layernorm_bwd_partials_bytes(int64_t rows, int C) {
    return (size_t)((rows + LN_ROWS_PER_WG - 1) / LN_ROWS_PER_WG) * 2 * C * sizeof(float);
}

// internal (block.hip): `partials` (maed_layernorm_bwd_partials_bytes, or null) selects the store-then-sum form of dgamma/dbeta
int maed_layernorm_bwd_ws(const void* dy, int dtype, const float* x, int64_t x_row_stride, const float* gamma,
                          const float* mean, const float* rstd, const float* dres_in, float* dx_out, void* dx_twin,
                          float* dgamma, float* dbeta, int64_t rows, int C, float* partials, void* stream, bool finish = true, uint32_t* clear = nullptr,
                          int clear_words = 0);
// the closing column sum of the store-then-sum form on its own (block.hip runs it on the side stream: finish = false above, then this)
int maed_layernorm_affine_finish(const float* partials, int64_t rows, int C, float* dgamma, float* dbeta, void* stream) {
    const int nwg = (int)((rows + LN_ROWS_PER_WG - 1) / LN_ROWS_PER_WG);
    hipLaunchKernelGGL(ln_affine_finish_kernel, dim3((2 * C + 63) / 64, (nwg + 63) / 64), dim3(256), 0, (hipStream_t)stream, partials, nwg, C, dgamma, dbeta);
    MAED_CHECK_LAUNCH("layernorm_affine_finish");
    return MAED_OK;
}

extern "C" int maed_layernorm_bwd(const void* dy, int dtype, const float* x, int64_t x_row_stride, const float* gamma,
                                  const float* mean, const float* rstd, const float* dres_in, float* dx_out, void* dx_twin,
                                  float* dgamma, float* dbeta, int64_t rows, int C, void* stream) {
    return maed_layernorm_bwd_ws(dy, dtype, x, x_row_stride, gamma, mean, rstd, dres_in, dx_out, dx_twin, dgamma, dbeta, rows, C, nullptr, stream);
}

int maed_layernorm_bwd_ws(const void* dy, int dtype, const float* x, int64_t x_row_stride, const float* gamma,
                          const float* mean, const float* rstd, const float* dres_in, float* dx_out, void* dx_twin,
                          float* dgamma, float* dbeta, int64_t rows, int C, float* partials, void* stream, bool finish, uint32_t* clear, int clear_words) {
    MAED_CHECK_ARG(dy && x && gamma && mean && rstd && dx_out && dgamma && dbeta, MAED_ERR_ARG, "layernorm_bwd: null pointer");
    MAED_CHECK_ARG(C > 0 && C % 4 == 0 && C <= 64 * 4 * LN_MAXV, MAED_ERR_SHAPE, "layernorm_bwd: C=%d unsupported", C);
    MAED_CHECK_ARG(x_row_stride % 4 == 0 && is_aligned(x, 16) && is_aligned(dy, 8) && is_aligned(dx_out, 16), MAED_ERR_ALIGN, "layernorm_bwd: alignment");
    if (rows == 0) return MAED_OK;
    dim3 grid((unsigned)((rows + LN_ROWS_PER_WG - 1) / LN_ROWS_PER_WG));
    MAED_CHECK_ARG(!clear || (int64_t)clear_words <= (int64_t)grid.x * 256, MAED_ERR_SHAPE, "layernorm_bwd: %d words to clear exceed the grid", clear_words);
    const size_t lds = (size_t)8 * C * sizeof(float);
#define LN_BWD(NV_) hipLaunchKernelGGL((ln_bwd_kernel<T, NV_>), grid, dim3(256), lds, (hipStream_t)stream, (const T*)dy, x, x_row_stride, gamma, mean, \
                                      rstd, dres_in, dx_out, (T*)dx_twin, dgamma, dbeta, rows, C, clear, clear_words)
#define LN_BWD_P(NV_) hipLaunchKernelGGL((ln_bwd_kernel<T, NV_, true>), grid, dim3(256), lds, (hipStream_t)stream, (const T*)dy, x, x_row_stride, gamma, \
                                        mean, rstd, dres_in, dx_out, (T*)dx_twin, partials, (float*)nullptr, rows, C, clear, clear_words)
    if (partials) {
        MAED_DISPATCH_DTYPE(dtype, T, { if (C <= 512) LN_BWD_P(2); else if (C <= 768) LN_BWD_P(3); else if (C <= 1024) LN_BWD_P(4); else LN_BWD_P(8); });
        if (finish) hipLaunchKernelGGL(ln_affine_finish_kernel, dim3((2 * C + 63) / 64, (grid.x + 63) / 64), dim3(256), 0, (hipStream_t)stream, partials, (int)grid.x, C,
                                       dgamma, dbeta);
    } else {
        MAED_DISPATCH_DTYPE(dtype, T, { if (C <= 512) LN_BWD(2); else if (C <= 768) LN_BWD(3); else if (C <= 1024) LN_BWD(4); else LN_BWD(8); });
    }
#undef LN_BWD_P
#undef LN_BWD
    MAED_CHECK_LAUNCH("layernorm_bwd");
    return MAED_OK;
}
