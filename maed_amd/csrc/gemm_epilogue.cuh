// Fused GEMM epilogues shared by the NT GEMM kernels (gemm.hip: 128x128 tiles, gemm256.hip: 256x256 pipelined tiles) and the
// implicit-GEMM convolution: bias, bias+GELU (+pre-activation), bias+residual (fp32), *GELU', tanh, +aux, fp32 atomics.
#pragma once
#include "common.cuh"

struct EpiArgs {
    const float* bias;
    void* out; int64_t ldo;
    void* out2;
    const void* aux; int64_t ldaux;
};

template <int EPI, typename T>
__device__ __forceinline__ void epilogue_store(const EpiArgs& e, int64_t r, int64_t c, float acc) {
    if constexpr (EPI == MAED_EPI_STORE) {
        stf((T*)e.out + r * e.ldo + c, acc + (e.bias ? e.bias[c] : 0.f));
    } else if constexpr (EPI == MAED_EPI_GELU) {
        const float pre = acc + (e.bias ? e.bias[c] : 0.f);
        stf((T*)e.out2 + r * e.ldo + c, pre);
        stf((T*)e.out + r * e.ldo + c, gelu_fwd<T>(round_to<T>(pre)));  // activation of the STORED (rounded) pre-activation
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
        stf((T*)e.out + r * e.ldo + c, ldf((const T*)e.aux + r * e.ldaux + c) + acc + (e.bias ? e.bias[c] : 0.f));
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
        st4((T*)e.out + r * e.ldo + c0, v);
    } else if constexpr (EPI == MAED_EPI_GELU) {
        st4((T*)e.out2 + r * e.ldo + c0, v);
        // activation of the STORED (rounded) pre-activation, as the backward sees it -- rounded in registers, not read back
        float a[4] = {gelu_fwd<T>(round_to<T>(v[0])), gelu_fwd<T>(round_to<T>(v[1])), gelu_fwd<T>(round_to<T>(v[2])), gelu_fwd<T>(round_to<T>(v[3]))};
        st4((T*)e.out + r * e.ldo + c0, a);
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
        st8((T*)e.out + r * e.ldo + c0, v);
    } else if constexpr (EPI == MAED_EPI_GELU) {
        st8((T*)e.out2 + r * e.ldo + c0, v);
        float a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = gelu_fwd<T>(round_to<T>(v[j]));   // activation of the STORED (rounded) pre-activation
        st8((T*)e.out + r * e.ldo + c0, a);
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
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] += v[j];
        st8((T*)e.out + r * e.ldo + c0, x);
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
