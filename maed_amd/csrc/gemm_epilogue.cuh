// Fused GEMM epilogues shared by the NT GEMM kernels (gemm.hip: 128x128 tiles, gemm256.hip: 256x256 pipelined tiles) and the
// implicit-GEMM convolution: bias, bias+GELU (+pre-activation), bias+residual (fp32), *GELU', tanh, +aux, fp32 atomics.
#pragma once
#include "common.cuh"

struct EpiArgs {
    const float* bias;
    void* out; int64_t ldo;
    void* out2;
    const void* aux; int64_t ldaux;
    // optional second output of a convolution GEMM (resnetv2.py:35-49: every convolution of the backbone feeds a GroupNorm(32)): the
    // statistics of the STORED (bf16-rounded) output, sums[n][32][2] (fp64, pre-zeroed by the caller) += (sum x, sum x^2) with n = row / gn_hw,
    // group = column / (N / 32) -- the separate statistics pass over the activation tensor disappears
    double* gn_sums = nullptr; int gn_hw = 0;
    // fp32 products that leave bf16 twins for a bf16 backward (maed_gemm_nt_twin, block.hip's twin forward; T = float only): `twin` = bf16 copy of what goes to `out`
    // (STORE: the result; GELU: the activation), same leading dimension; out2_bf16: the GELU pre-activation (out2) is stored as bf16 ONLY -- nothing of the forward
    // chain reads it
    void* twin = nullptr; bool out2_bf16 = false;
    // plane storage of the same output (csrc/gemm_x3p.hip; T = float only): `twin` is the hi plane, `lo` = bf16(value - hi) with the twin's leading dimension -- the next
    // split product reads (twin, lo) instead of fp32; `out` may then be NULL (nothing else reads the fp32 form)
    void* lo = nullptr;
};

// the (hi, lo) bf16 planes of eight fp32 values: hi = bf16(v) (round to nearest even), lo = bf16(v - hi) -- gemm_x3.h's split4<2>
__device__ __forceinline__ void epi_store_planes8(bf16* hi, bf16* lo, const float (&v)[8]) {
    float l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) l[j] = v[j] - round_to<bf16>(v[j]);
    st8_nt(hi, v);
    if (lo) st8_nt(lo, l);
}

template <int EPI, typename T>
__device__ __forceinline__ void epilogue_store(const EpiArgs& e, int64_t r, int64_t c, float acc) {
    if constexpr (EPI == MAED_EPI_STORE) {
        const float v = acc + (e.bias ? e.bias[c] : 0.f);
        if (sizeof(T) != 4 || e.out) stf((T*)e.out + r * e.ldo + c, v);
        if constexpr (sizeof(T) == 4) { if (e.twin) stf((bf16*)e.twin + r * e.ldo + c, v); if (e.lo) stf((bf16*)e.lo + r * e.ldo + c, v - round_to<bf16>(v)); }
    } else if constexpr (EPI == MAED_EPI_GELU) {
        const float pre = acc + (e.bias ? e.bias[c] : 0.f);
        if (e.out2) { if (sizeof(T) == 4 && e.out2_bf16) stf((bf16*)e.out2 + r * e.ldo + c, pre); else stf((T*)e.out2 + r * e.ldo + c, pre); }   // (out2 = NULL: nobody will ask for GELU', inference)
        const float a = gelu_fwd<T>(round_to<T>(pre));
        if (sizeof(T) != 4 || e.out) stf((T*)e.out + r * e.ldo + c, a);  // activation of the STORED (rounded) pre-activation
        if constexpr (sizeof(T) == 4) { if (e.twin) stf((bf16*)e.twin + r * e.ldo + c, a); if (e.lo) stf((bf16*)e.lo + r * e.ldo + c, a - round_to<bf16>(a)); }
    } else if constexpr (EPI == MAED_EPI_RESID_F32) {
        ((float*)e.out)[r * e.ldo + c] = ((const float*)e.aux)[r * e.ldaux + c] + (acc + (e.bias ? e.bias[c] : 0.f));
    } else if constexpr (EPI == MAED_EPI_MUL_DGELU) {
        stf((T*)e.out + r * e.ldo + c, acc * gelu_bwd<T>(ldf((const T*)e.aux + r * e.ldaux + c)));
    } else if constexpr (EPI == MAED_EPI_ATOMIC_F32) {
        atomicAdd((float*)e.out + r * e.ldo + c, acc);
    } else if constexpr (EPI == MAED_EPI_STORE_F32) {
        ((float*)e.out)[r * e.ldo + c] = acc + (e.bias ? e.bias[c] : 0.f);
    } else if constexpr (EPI == MAED_EPI_TANH) {
        stf((T*)e.out + r * e.ldo + c, tanhf(acc + (e.bias ? e.bias[c] : 0.f)));
    } else if constexpr (EPI == MAED_EPI_ADD) {
        float x = ldf((const T*)e.aux + r * e.ldaux + c);
        if (e.out2) x = ((((const uint8_t*)e.out2)[(r * e.ldaux + c) >> 3] >> (c & 7)) & 1) ? x : 0.f;     // aux masked by 1 bit per element (see maed_gemm_nt)
        stf((T*)e.out + r * e.ldo + c, x + acc + (e.bias ? e.bias[c] : 0.f));
    }
}

// four consecutive columns c0..c0+3 of one row (c0 % 4 == 0): 8/16-byte accesses when the row base is aligned
template <int EPI, typename T>
__device__ __forceinline__ void epilogue_store4(const EpiArgs& e, int64_t r, int64_t c0, int64_t N, const float (&acc)[4], bool vec_ok) {
    if (!(vec_ok && c0 + 4 <= N)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (c0 + j < N) epilogue_store<EPI, T>(e, r, c0 + j, acc[j]);
        return;
    }
    float v[4] = {acc[0], acc[1], acc[2], acc[3]};
    if constexpr (EPI == MAED_EPI_ATOMIC_F32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd((float*)e.out + r * e.ldo + c0 + j, v[j]);
        return;
    }
    if constexpr (EPI != MAED_EPI_MUL_DGELU) {
        if (e.bias) { float b[4]; ld4(e.bias + c0, b); v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3]; }
    }
    if constexpr (EPI == MAED_EPI_STORE) {
        if (sizeof(T) != 4 || e.out) st4((T*)e.out + r * e.ldo + c0, v);
        if constexpr (sizeof(T) == 4) { if (e.twin) st4((bf16*)e.twin + r * e.ldo + c0, v); }
    } else if constexpr (EPI == MAED_EPI_GELU) {
        if (e.out2) { if (sizeof(T) == 4 && e.out2_bf16) st4((bf16*)e.out2 + r * e.ldo + c0, v); else st4((T*)e.out2 + r * e.ldo + c0, v); }
        // activation of the STORED (rounded) pre-activation, as the backward sees it -- rounded in registers, not read back
        float a[4] = {gelu_fwd<T>(round_to<T>(v[0])), gelu_fwd<T>(round_to<T>(v[1])), gelu_fwd<T>(round_to<T>(v[2])), gelu_fwd<T>(round_to<T>(v[3]))};
        if (sizeof(T) != 4 || e.out) st4((T*)e.out + r * e.ldo + c0, a);
        if constexpr (sizeof(T) == 4) { if (e.twin) st4((bf16*)e.twin + r * e.ldo + c0, a); }
    } else if constexpr (EPI == MAED_EPI_RESID_F32) {
        float x[4]; ld4((const float*)e.aux + r * e.ldaux + c0, x);
        float o[4] = {x[0] + v[0], x[1] + v[1], x[2] + v[2], x[3] + v[3]};
        st4((float*)e.out + r * e.ldo + c0, o);
    } else if constexpr (EPI == MAED_EPI_MUL_DGELU) {
        float x[4]; ld4((const T*)e.aux + r * e.ldaux + c0, x);
        float o[4] = {v[0] * gelu_bwd<T>(x[0]), v[1] * gelu_bwd<T>(x[1]), v[2] * gelu_bwd<T>(x[2]), v[3] * gelu_bwd<T>(x[3])};
        st4((T*)e.out + r * e.ldo + c0, o);
    } else if constexpr (EPI == MAED_EPI_STORE_F32) {
        st4((float*)e.out + r * e.ldo + c0, v);
    } else if constexpr (EPI == MAED_EPI_TANH) {
        float o[4] = {tanhf(v[0]), tanhf(v[1]), tanhf(v[2]), tanhf(v[3])};
        st4((T*)e.out + r * e.ldo + c0, o);
    } else if constexpr (EPI == MAED_EPI_ADD) {
        float x[4]; ld4((const T*)e.aux + r * e.ldaux + c0, x);
        if (e.out2) {
            const uint32_t m = ((const uint8_t*)e.out2)[(r * e.ldaux + c0) >> 3] >> (c0 & 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = ((m >> j) & 1u) ? x[j] : 0.f;
        }
        float o[4] = {x[0] + v[0], x[1] + v[1], x[2] + v[2], x[3] + v[3]};
        st4((T*)e.out + r * e.ldo + c0, o);
    }
}

// eight consecutive columns c0..c0+7 of one row (c0 % 8 == 0): 16/32-byte accesses (the LDS-shuffled epilogue of the
// direct-to-LDS kernels: 8 lanes cover a 64-column row segment = one or two full cache lines per row)
template <int EPI, typename T>
__device__ __forceinline__ void epilogue_store8(const EpiArgs& e, int64_t r, int64_t c0, int64_t N, const float (&acc)[8], bool vec_ok) {
    if (!(vec_ok && c0 + 8 <= N)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) if (c0 + j < N) epilogue_store<EPI, T>(e, r, c0 + j, acc[j]);
        return;
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = acc[j];
    if constexpr (EPI != MAED_EPI_MUL_DGELU) {
        if (e.bias) { float b[8]; ld8(e.bias + c0, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += b[j]; }
    }
    if constexpr (EPI == MAED_EPI_STORE) {
        if (sizeof(T) != 4 || e.out) st8((T*)e.out + r * e.ldo + c0, v);
        if constexpr (sizeof(T) == 4) { if (e.twin) epi_store_planes8((bf16*)e.twin + r * e.ldo + c0, e.lo ? (bf16*)e.lo + r * e.ldo + c0 : nullptr, v); }
    } else if constexpr (EPI == MAED_EPI_GELU) {
        if (e.out2) { if (sizeof(T) == 4 && e.out2_bf16) st8_nt((bf16*)e.out2 + r * e.ldo + c0, v); else st8((T*)e.out2 + r * e.ldo + c0, v); }
        float a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = gelu_fwd<T>(round_to<T>(v[j]));   // activation of the STORED (rounded) pre-activation
        if (sizeof(T) != 4 || e.out) st8((T*)e.out + r * e.ldo + c0, a);
        if constexpr (sizeof(T) == 4) { if (e.twin) epi_store_planes8((bf16*)e.twin + r * e.ldo + c0, e.lo ? (bf16*)e.lo + r * e.ldo + c0 : nullptr, a); }
    } else if constexpr (EPI == MAED_EPI_RESID_F32) {
        float x[8]; ld8((const float*)e.aux + r * e.ldaux + c0, x);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] += v[j];
        st8((float*)e.out + r * e.ldo + c0, x);
    } else if constexpr (EPI == MAED_EPI_MUL_DGELU) {
        float x[8]; ld8((const T*)e.aux + r * e.ldaux + c0, x);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = v[j] * gelu_bwd<T>(x[j]);
        st8((T*)e.out + r * e.ldo + c0, x);
    } else if constexpr (EPI == MAED_EPI_STORE_F32) {
        st8((float*)e.out + r * e.ldo + c0, v);
    } else if constexpr (EPI == MAED_EPI_TANH) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tanhf(v[j]);
        st8((T*)e.out + r * e.ldo + c0, v);
    } else if constexpr (EPI == MAED_EPI_ADD) {
        float x[8]; ld8((const T*)e.aux + r * e.ldaux + c0, x);
        if (e.out2) {
            const uint32_t m = ((const uint8_t*)e.out2)[(r * e.ldaux + c0) >> 3];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = ((m >> j) & 1u) ? x[j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] += v[j];
        st8((T*)e.out + r * e.ldo + c0, x);
    }
}

// ---- GroupNorm statistics in the LDS-shuffled epilogue of a 128-row tile ------------------------------------------------------------
// A lane of the shuffled epilogue owns 8 consecutive output columns (c0 fixed) of 8 rows: it keeps (sum, sum of squares) per column
// PAIR (the finest group the backbone has is 2 channels) and per frame (a tile of 128 rows touches at most two frames when
// gn_hw >= 128) in registers, the 8 lanes of a wave that share c0 are folded with three shuffles, and only then 8 lanes per wave
// add into the workgroup's LDS table [2 frames][64 groups][2] floats (zeroed at kernel start); gn_flush adds the table to the global
// fp64 sums -- (2 frames x groups x 2) fp64 atomics per workgroup.
// (fp64 table: the order in which waves arrive then changes the sums by ~1e-16 relative, never a bf16 rounding of the normalised output -- with fp32
// LDS atomics the statistics moved in the 7th digit from run to run, and 50 layers of bf16 rounding amplify a flipped ulp chaotically)
#define GN_TAB_FLOATS (2 * 64 * 2 * 2)                                      /* table size in 4-byte units: [2 frames][64 groups][2] doubles */
struct GnTile { double* tab; int64_t split_row; int sh, g0; };            // split_row: first row of the tile's second frame; sh = log2(channels per group)
struct GnRegs { float s[2][4], q[2][4]; };
__device__ __forceinline__ GnTile gn_tile(double* tab, int64_t m0, int64_t n0, int64_t N, int hw) {
    const int cpg = (int)(N >> 5);                                        // N % 32 == 0 and a power of two per group (host-checked)
    const int sh = 31 - __builtin_clz((unsigned)cpg);
    return GnTile{tab, (m0 / hw + 1) * (int64_t)hw, sh, (int)(n0 >> sh)};
}
__device__ __forceinline__ void gn_zero(GnRegs& a) {
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int h = 0; h < 4; ++h) { a.s[f][h] = 0.f; a.q[f][h] = 0.f; }
}
template <typename T = bf16>                                             // T: storage type of the convolution's output
__device__ __forceinline__ void gn_acc8(GnRegs& a, const GnTile& g, const float (&v)[8], int64_t row) {
    const bool second = row >= g.split_row;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const float r0 = round_to<T>(v[2 * h]), r1 = round_to<T>(v[2 * h + 1]);            // what the GroupNorm will read back
        const float ps = r0 + r1, pq = fmaf(r0, r0, r1 * r1);
        a.s[0][h] += second ? 0.f : ps; a.q[0][h] += second ? 0.f : pq;
        a.s[1][h] += second ? ps : 0.f; a.q[1][h] += second ? pq : 0.f;
    }
}
// after the last gn_acc8: fold the lanes that share c0 (lane bits 3..5) and add to the LDS table
__device__ __forceinline__ void gn_commit(GnRegs& a, const GnTile& g, int lane, int64_t c0, int64_t N) {
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
#pragma unroll
            for (int m = 8; m < 64; m <<= 1) { a.s[f][h] += __shfl_xor(a.s[f][h], m); a.q[f][h] += __shfl_xor(a.q[f][h], m); }
        }
    if ((lane >> 3) != 0 || c0 >= N) return;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        double* const tf = g.tab + f * 128;
        if (g.sh >= 3) {                                                  // >= 8 channels per group: the lane's 8 columns are one group
            double* t = tf + ((((int)(c0 >> g.sh)) - g.g0) << 1);
            atomicAdd(t, (double)((a.s[f][0] + a.s[f][1]) + (a.s[f][2] + a.s[f][3]))); atomicAdd(t + 1, (double)((a.q[f][0] + a.q[f][1]) + (a.q[f][2] + a.q[f][3])));
        } else if (g.sh == 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double* t = tf + ((((int)(c0 >> 2)) + h - g.g0) << 1);
                atomicAdd(t, (double)(a.s[f][2 * h] + a.s[f][2 * h + 1])); atomicAdd(t + 1, (double)(a.q[f][2 * h] + a.q[f][2 * h + 1]));
            }
        } else {                                                          // 2 channels per group
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                double* t = tf + ((((int)(c0 >> 1)) + h - g.g0) << 1);
                atomicAdd(t, (double)a.s[f][h]); atomicAdd(t + 1, (double)a.q[f][h]);
            }
        }
    }
}
__device__ __forceinline__ void gn_flush(const GnTile& g, double* sums, int64_t m0, int64_t M, int hw, int tile_cols, int tid, int nthr) {
    const int ngl = ((tile_cols - 1) >> g.sh) + 1;                        // groups this tile's columns touch (<= 64)
    const int64_t n_first = m0 / hw;
    for (int i = tid; i < 2 * ngl * 2; i += nthr) {
        const int k = i & 1, gl = (i >> 1) % ngl, f = (i >> 1) / ngl;
        const int64_t n = n_first + f;
        const int grp = g.g0 + gl;
        const double v = g.tab[f * 128 + (gl << 1) + k];
        if (n * hw < M && grp < 32 && v != 0.0) atomicAdd(sums + (n * 32 + grp) * 2 + k, v);
    }
}


__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // bijective remap: blocks that the dispatcher places on XCD x (bid % 8 == x) get a contiguous id range
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

#define GL_ST 68   // fp32 row stride of the epilogue staging area: 272 B -> conflict-free ds_write_b128 per 16-lane group
typedef maed_lds_void_t lds_void_t;
typedef maed_glb_void_t glb_void_t;
