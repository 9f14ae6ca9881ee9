// K3: spatial attention of the STE (vision_transformer.py:206-214): per (frame, head)
//     o = softmax(q k^T * scale) v  over the P tokens of one frame, d = 64.
// The reference materialises the (F,H,P,P) score tensor twice in HBM; here scores never leave
// registers, K/V of the head are staged once in LDS, and q/k/v are read straight out of the qkv
// Linear's (F,P,3C) output (no permuted copies).
//
//  * bf16 forward (MFMA): one workgroup per (frame, head), one wave per 32-row q tile.
//    S^T = K Q^T via v_mfma_f32_32x32x16_bf16 (A = K rows from LDS, B = Q rows in registers), so a
//    lane owns one q column and 16 keys of it: the softmax row reduction is in-register plus one
//    cross-half shuffle.  O^T = V^T P^T uses the exponentiated scores directly as the B fragment
//    (k-slot permutation shared with the V^T fragment read), so P never touches LDS and the
//    online-softmax rescale of O^T is lane-local.  V is transposed once while staging.
//  * generic-T VALU kernels (f32 parity mode, and the backward of both modes in this round):
//    thread per row, the other side's matrices broadcast-read from LDS.
#include "common.cuh"
#include "gemm_x3.h"

#define D HEAD_DIM

// ==================================================================================================
// VALU forward / backward (generic T)
// ==================================================================================================
template <typename T>
__device__ __forceinline__ void stage_rows_f32(float* dst, int ldd, const T* src, int64_t row_stride, int P) {
    // dst[p][0..63] (row stride ldd floats) <- src[p*row_stride + 0..63]
    for (int idx = threadIdx.x; idx < P * (D / 4); idx += blockDim.x) {
        const int p = idx / (D / 4), c = (idx % (D / 4)) * 4;
        float v[4];
        ld4(src + (int64_t)p * row_stride + c, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[p * ldd + c + j] = v[j];
    }
}

#define LDF 65  // padded LDS row (floats)

template <typename T>
__global__ __launch_bounds__(256) void attn_sp_fwd_valu(const T* __restrict__ qkv, T* __restrict__ o, float* __restrict__ lse,
                                                        int P, int H, float scale) {
    MAED_DYN_SHARED(float, sm);
    float* Ks = sm; float* Vs = sm + (size_t)P * LDF;
    const int f = blockIdx.x / H, h = blockIdx.x % H;
    const int C = H * D;
    const int64_t ld = 3 * (int64_t)C;
    const T* base = qkv + (int64_t)f * P * ld + h * D;
    stage_rows_f32(Ks, LDF, base + C, ld, P);
    stage_rows_f32(Vs, LDF, base + 2 * C, ld, P);
    __syncthreads();
    for (int q = threadIdx.x; q < P; q += blockDim.x) {
        float qv[D], acc[D];
#pragma unroll
        for (int c = 0; c < D; c += 4) { float v[4]; ld4(base + (int64_t)q * ld + c, v); qv[c] = v[0]; qv[c + 1] = v[1]; qv[c + 2] = v[2]; qv[c + 3] = v[3]; }
#pragma unroll
        for (int c = 0; c < D; ++c) acc[c] = 0.f;
        float m = -INFINITY, l = 0.f;
        for (int k = 0; k < P; ++k) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) s = fmaf(qv[c], Ks[k * LDF + c], s);
            s *= scale;
            const float mn = fmaxf(m, s);
            const float a = __expf(m - mn), p = __expf(s - mn);
            l = l * a + p;
#pragma unroll
            for (int c = 0; c < D; ++c) acc[c] = fmaf(p, Vs[k * LDF + c], acc[c] * a);
            m = mn;
        }
        const float inv = 1.f / l;
        T* orow = o + ((int64_t)f * P + q) * C + h * D;
#pragma unroll
        for (int c = 0; c < D; c += 4) { float v[4] = {acc[c] * inv, acc[c + 1] * inv, acc[c + 2] * inv, acc[c + 3] * inv}; st4(orow + c, v); }
        lse[((int64_t)f * H + h) * P + q] = m + __logf(l);
    }
}

// dQ: thread per query row, K/V of the head in LDS
template <typename T>
__global__ __launch_bounds__(256) void attn_sp_bwd_dq_valu(const T* __restrict__ qkv, const T* __restrict__ o,
                                                           const T* __restrict__ d_o, const float* __restrict__ lse,
                                                           T* __restrict__ dqkv, int accumulate, int P, int H, float scale) {
    MAED_DYN_SHARED(float, sm);
    float* Ks = sm; float* Vs = sm + (size_t)P * LDF;
    const int f = blockIdx.x / H, h = blockIdx.x % H;
    const int C = H * D;
    const int64_t ld = 3 * (int64_t)C;
    const T* base = qkv + (int64_t)f * P * ld + h * D;
    stage_rows_f32(Ks, LDF, base + C, ld, P);
    stage_rows_f32(Vs, LDF, base + 2 * C, ld, P);
    __syncthreads();
    for (int q = threadIdx.x; q < P; q += blockDim.x) {
        float qv[D], dov[D], dq[D];
        const T* orow = o + ((int64_t)f * P + q) * C + h * D;
        const T* dorow = d_o + ((int64_t)f * P + q) * C + h * D;
        float Dq = 0.f;
#pragma unroll
        for (int c = 0; c < D; c += 4) {
            float v[4], w[4], u[4];
            ld4(base + (int64_t)q * ld + c, v); ld4(dorow + c, w); ld4(orow + c, u);
#pragma unroll
            for (int j = 0; j < 4; ++j) { qv[c + j] = v[j]; dov[c + j] = w[j]; Dq = fmaf(w[j], u[j], Dq); dq[c + j] = 0.f; }
        }
        const float L = lse[((int64_t)f * H + h) * P + q];
        for (int k = 0; k < P; ++k) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) { s = fmaf(qv[c], Ks[k * LDF + c], s); dp = fmaf(dov[c], Vs[k * LDF + c], dp); }
            const float p = __expf(s * scale - L);
            const float ds = p * (dp - Dq) * scale;
#pragma unroll
            for (int c = 0; c < D; ++c) dq[c] = fmaf(ds, Ks[k * LDF + c], dq[c]);
        }
        T* dst = dqkv + ((int64_t)f * P + q) * ld + h * D;
#pragma unroll
        for (int c = 0; c < D; c += 4) {
            float v[4] = {dq[c], dq[c + 1], dq[c + 2], dq[c + 3]};
            if (accumulate) { float old[4]; ld4(dst + c, old); v[0] += old[0]; v[1] += old[1]; v[2] += old[2]; v[3] += old[3]; }
            st4(dst + c, v);
        }
    }
}

// dK, dV: thread per key row, Q/dO of the head (+ lse, D) in LDS; two sweeps to stay in registers
template <typename T>
__global__ __launch_bounds__(256) void attn_sp_bwd_dkv_valu(const T* __restrict__ qkv, const T* __restrict__ o,
                                                            const T* __restrict__ d_o, const float* __restrict__ lse,
                                                            T* __restrict__ dqkv, int accumulate, int P, int H, float scale) {
    MAED_DYN_SHARED(float, sm);
    float* Qs = sm; float* dOs = sm + (size_t)P * LDF; float* Ls = dOs + (size_t)P * LDF; float* Ds = Ls + P;
    const int f = blockIdx.x / H, h = blockIdx.x % H;
    const int C = H * D;
    const int64_t ld = 3 * (int64_t)C;
    const T* base = qkv + (int64_t)f * P * ld + h * D;
    const T* obase = o + (int64_t)f * P * C + h * D;
    const T* dobase = d_o + (int64_t)f * P * C + h * D;
    stage_rows_f32(Qs, LDF, base, ld, P);
    stage_rows_f32(dOs, LDF, dobase, C, P);
    __syncthreads();
    for (int q = threadIdx.x; q < P; q += blockDim.x) {
        float Dq = 0.f;
#pragma unroll
        for (int c = 0; c < D; c += 4) { float u[4]; ld4(obase + (int64_t)q * C + c, u);
#pragma unroll
            for (int j = 0; j < 4; ++j) Dq = fmaf(dOs[q * LDF + c + j], u[j], Dq); }
        Ds[q] = Dq;
        Ls[q] = lse[((int64_t)f * H + h) * P + q];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < P; k += blockDim.x) {
        float kv[D], acc[D];
#pragma unroll
        for (int c = 0; c < D; c += 4) { float v[4]; ld4(base + C + (int64_t)k * ld + c, v); kv[c] = v[0]; kv[c + 1] = v[1]; kv[c + 2] = v[2]; kv[c + 3] = v[3]; }
        // sweep 1: dV[k] = sum_q p[q][k] dO[q]
#pragma unroll
        for (int c = 0; c < D; ++c) acc[c] = 0.f;
        for (int q = 0; q < P; ++q) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) s = fmaf(Qs[q * LDF + c], kv[c], s);
            const float p = __expf(s * scale - Ls[q]);
#pragma unroll
            for (int c = 0; c < D; ++c) acc[c] = fmaf(p, dOs[q * LDF + c], acc[c]);
        }
        T* dv = dqkv + ((int64_t)f * P + k) * ld + 2 * C + h * D;
#pragma unroll
        for (int c = 0; c < D; c += 4) {
            float v[4] = {acc[c], acc[c + 1], acc[c + 2], acc[c + 3]};
            if (accumulate) { float old[4]; ld4(dv + c, old); v[0] += old[0]; v[1] += old[1]; v[2] += old[2]; v[3] += old[3]; }
            st4(dv + c, v);
        }
        // sweep 2: dK[k] = sum_q ds[q][k] Q[q]
        float vv[D];
#pragma unroll
        for (int c = 0; c < D; c += 4) { float v[4]; ld4(base + 2 * C + (int64_t)k * ld + c, v); vv[c] = v[0]; vv[c + 1] = v[1]; vv[c + 2] = v[2]; vv[c + 3] = v[3]; }
#pragma unroll
        for (int c = 0; c < D; ++c) acc[c] = 0.f;
        for (int q = 0; q < P; ++q) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) { s = fmaf(Qs[q * LDF + c], kv[c], s); dp = fmaf(dOs[q * LDF + c], vv[c], dp); }
            const float p = __expf(s * scale - Ls[q]);
            const float ds = p * (dp - Ds[q]) * scale;
#pragma unroll
            for (int c = 0; c < D; ++c) acc[c] = fmaf(ds, Qs[q * LDF + c], acc[c]);
        }
        T* dk = dqkv + ((int64_t)f * P + k) * ld + C + h * D;
#pragma unroll
        for (int c = 0; c < D; c += 4) {
            float v[4] = {acc[c], acc[c + 1], acc[c + 2], acc[c + 3]};
            if (accumulate) { float old[4]; ld4(dk + c, old); v[0] += old[0]; v[1] += old[1]; v[2] += old[2]; v[3] += old[3]; }
            st4(dk + c, v);
        }
    }
}

// blocks b, b+8, b+16, ... share an XCD (observed dispatch: XCD = b % 8).  Remap so that one XCD walks CONTIGUOUS
// (frame, head) pairs: the 8 heads of a frame (adjacent 128-B lines of every qkv row) then meet in ONE L2 close in time
// instead of being fetched piecemeal by 8 different XCDs.  Pure speed heuristic; any placement is correct.
__device__ __forceinline__ int attn_xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// ==================================================================================================
// MFMA forward (bf16)
// ==================================================================================================
#include "attn_mfma.cuh"

__global__ __launch_bounds__(1024) void attn_sp_fwd_mfma(const bf16* __restrict__ qkv, bf16* __restrict__ o,
                                                         float* __restrict__ lse, int P, int H, float scale_log2e) {
    MAED_DYN_SHARED(unsigned short, smem);
    const int Pk = (P + 31) & ~31;
    const int VLD = Pk + 4;                        // V^T row stride (elements): (Pk+4)/2 dwords = 2*odd -> conflict-free b64
    // K rows are stored UNPADDED (128 B) with the 16-B chunk index XOR-swizzled by (row>>1)&7: a ds_read_b128 of 16
    // distinct rows (mod 16) at one logical chunk then covers all 16 slots of the 256-B bank row -> conflict-free, and
    // the workgroup's LDS drops to 54.4 KB (P=197), i.e. three workgroups per CU instead of two.
    unsigned short* Ks = smem;                     // [P][64] swizzled
    unsigned short* Vt = smem + (size_t)P * 64;    // [64][VLD]
    const int bid = attn_xcd_remap(blockIdx.x, gridDim.x);
    const int f = bid / H, h = bid % H;
    const int C = H * D;
    const int64_t ld = 3 * (int64_t)C;
    const bf16* base = qkv + (int64_t)f * P * ld + h * D;
    const int tid = threadIdx.x, nthr = blockDim.x;

    // this wave's 32 query rows: fragment loads issued first so they fly together with the K/V staging loads
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int q0 = wave * 32;
    const int q = q0 + l31;
    const int qc = q < P ? q : P - 1;
    bf16x8_t qf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) qf[t] = *reinterpret_cast<const bf16x8_t*>(base + (int64_t)qc * ld + t * 16 + hi * 8);

    // ---- stage K (row-major, padded) and V^T (transposed while writing) -----------------------------
    // blockDim = 64*ceil(P/32) >= 2P threads and P*8 chunks => at most 4 chunks per thread: issue ALL global
    // loads first (8 x 16 B in flight per lane), then write LDS -- one memory latency instead of four.
    {
        uint4 kreg[4], vreg[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int idx = tid + i * nthr;
            if (idx > P * 8 - 1) idx = P * 8 - 1;
            const int p = idx >> 3, c8 = (idx & 7) * 8;
            kreg[i] = *reinterpret_cast<const uint4*>(base + C + (int64_t)p * ld + c8);
            vreg[i] = *reinterpret_cast<const uint4*>(base + 2 * C + (int64_t)p * ld + c8);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * nthr;
            if (idx < P * 8) {
                const int p = idx >> 3, c8 = (idx & 7) * 8;
                *reinterpret_cast<uint4*>(Ks + p * 64 + (((c8 >> 3) ^ ((p >> 1) & 7)) << 3)) = kreg[i];
                const uint32_t w[4] = {vreg[i].x, vreg[i].y, vreg[i].z, vreg[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    Vt[(c8 + 2 * j) * VLD + p] = (unsigned short)(w[j] & 0xffffu);
                    Vt[(c8 + 2 * j + 1) * VLD + p] = (unsigned short)(w[j] >> 16);
                }
            }
        }
    }
    // zero the key padding of V^T (columns P .. VLD-1, at most 35): shifts only, no runtime integer division
    for (int i = tid; i < D * 64; i += nthr) {
        const int e = i >> 6, j = i & 63;
        if (P + j < VLD) Vt[e * VLD + P + j] = 0;
    }
    __syncthreads();

    if (q0 >= P) return;

    f32x16_t oacc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; }
    float m = -INFINITY, l = 0.f;

    const int nkt = Pk / 32;
    for (int kt = 0; kt < nkt; ++kt) {
        // S^T tile: rows = keys kt*32.., cols = this wave's 32 queries
        f32x16_t s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        int krow = kt * 32 + l31;
        if (krow > P - 1) krow = P - 1;
        const unsigned short* kp = Ks + krow * 64;
        const int swz = (krow >> 1) & 7;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(kp + (((2 * t + hi) ^ swz) << 3));
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[t], s, 0, 0, 0);
        }
        // lane holds keys k(r) = kt*32 + (r&3) + 8*(r>>2) + 4*hi of query q; only the last tile has padding keys
        if (kt == nkt - 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                s[r] = (k < P) ? s[r] : -INFINITY;
            }
        }
        float mt = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * scale_log2e;       // scale > 0: max commutes with the scaling
        const float mn = fmaxf(m, mt);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, -mn)); ps += s[r]; }
        l = l * alpha + ps;
        m = mn;
#pragma unroll
        for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }
        // O^T += V^T P^T : k-step st of 16 keys uses regs 8*st .. 8*st+7 (k = kt*32 + 16*st + 8*(j>>2) + 4*hi + (j&3))
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
    }
    l += __shfl_xor(l, 32, 64);
    if (q < P) {
        const float inv = 1.f / l;
        bf16* orow = o + ((int64_t)f * P + q) * C + h * D;
        // O^T tile et: lane holds e = et*32 + (r&3) + 8*(r>>2) + 4*hi for its query
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int e0 = et * 32 + 8 * g + 4 * hi;
                const uint2 w = make_uint2(pack_bf2(oacc[et][4 * g] * inv, oacc[et][4 * g + 1] * inv),
                                           pack_bf2(oacc[et][4 * g + 2] * inv, oacc[et][4 * g + 3] * inv));
                *reinterpret_cast<uint2*>(orow + e0) = w;
            }
        if (hi == 0) lse[((int64_t)f * H + h) * P + q] = (m + log2f(l)) * 0.69314718055994530942f;
    }
}

// ==================================================================================================
// MFMA backward (bf16): flash-style recompute from the saved log-sum-exp, two kernels, whole head in LDS.
//   dQ pass : wave per 32 queries.  S^T = K Q^T and dP^T = V dO^T (lane = query, 16 keys per tile in
//             registers) -> dS in registers -> dQ^T += K^T dS^T with dS as the MFMA B fragment (no LDS
//             round trip, same k-slot trick as the forward) and K^T read from a transposed LDS image.
//   dK/dV   : wave per 32 keys.  S = Q K^T and dP = dO V^T (lane = key, 16 queries per tile) ->
//             dV^T += dO^T P, dK^T += Q^T dS with P / dS as B fragments, Q^T / dO^T transposed in LDS.
// ==================================================================================================
__device__ __forceinline__ void stage_rowmajor_and_transposed(unsigned short* rows, unsigned short* tr, int VLD,
                                                              const bf16* src, int64_t ld, int P, int tid, int nthr) {
    // <= 4 chunks of 16 B per thread (blockDim >= 2P): all loads in flight first, then the LDS writes
    uint4 reg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int idx = tid + i * nthr;
        if (idx > P * 8 - 1) idx = P * 8 - 1;
        reg[i] = *reinterpret_cast<const uint4*>(src + (int64_t)(idx >> 3) * ld + (idx & 7) * 8);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * nthr;
        if (idx < P * 8) {
            const int p = idx >> 3, c8 = (idx & 7) * 8;
            if (rows) *reinterpret_cast<uint4*>(rows + p * KLD + c8) = reg[i];
            if (tr) {
                const uint32_t w[4] = {reg[i].x, reg[i].y, reg[i].z, reg[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    tr[(c8 + 2 * j) * VLD + p] = (unsigned short)(w[j] & 0xffffu);
                    tr[(c8 + 2 * j + 1) * VLD + p] = (unsigned short)(w[j] >> 16);
                }
            }
        }
    }
    if (tr)
        for (int i = tid; i < D * 64; i += nthr) {   // zero columns P .. VLD-1 (<= 35): no runtime integer division
            const int e = i >> 6, j = i & 63;
            if (P + j < VLD) tr[e * VLD + P + j] = 0;
        }
}

__global__ __launch_bounds__(640) void attn_sp_bwd_dq_mfma(const bf16* __restrict__ qkv, const bf16* __restrict__ o,
                                                            const bf16* __restrict__ d_o, const float* __restrict__ lse,
                                                            bf16* __restrict__ dqkv, int accumulate, int P, int H, float scale) {
    MAED_DYN_SHARED(unsigned short, smem);
    const int Pk = (P + 31) & ~31, VLD = Pk + 4;
    unsigned short* Ks = smem;
    unsigned short* Vs = Ks + (size_t)P * KLD;
    unsigned short* Kt = Vs + (size_t)P * KLD;
    const int bid = attn_xcd_remap(blockIdx.x, gridDim.x);
    const int f = bid / H, h = bid % H, C = H * D;
    const int64_t ld = 3 * (int64_t)C;
    const bf16* base = qkv + (int64_t)f * P * ld + h * D;
    const int tid = threadIdx.x, nthr = blockDim.x;
    stage_rowmajor_and_transposed(Ks, Kt, VLD, base + C, ld, P, tid, nthr);
    stage_rowmajor_and_transposed(Vs, nullptr, VLD, base + 2 * C, ld, P, tid, nthr);
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int q = wave * 32 + l31;
    if (wave * 32 >= P) return;
    const int qc = q < P ? q : P - 1;
    const bf16* orow = o + ((int64_t)f * P + qc) * C + h * D;
    const bf16* dorow = d_o + ((int64_t)f * P + qc) * C + h * D;
    bf16x8_t qf[4], dof[4];
    float Dq = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        qf[t] = *reinterpret_cast<const bf16x8_t*>(base + (int64_t)qc * ld + t * 16 + hi * 8);
        dof[t] = *reinterpret_cast<const bf16x8_t*>(dorow + t * 16 + hi * 8);
        float a[8], b[8];
        ld8(dorow + t * 16 + hi * 8, a); ld8(orow + t * 16 + hi * 8, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) Dq = fmaf(a[j], b[j], Dq);
    }
    Dq += __shfl_xor(Dq, 32, 64);
    const float l2e = 1.44269504088896340736f;
    const float L2 = lse[((int64_t)f * H + h) * P + qc] * l2e, sl2e = scale * l2e;
    f32x16_t dq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }
    const int nkt = Pk / 32;
    for (int kt = 0; kt < nkt; ++kt) {
        f32x16_t s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        int krow = kt * 32 + l31;
        if (krow > P - 1) krow = P - 1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + krow * KLD + t * 16 + hi * 8);
            const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(Vs + krow * KLD + t * 16 + hi * 8);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[t], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[t], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const float p = (k < P) ? __builtin_amdgcn_exp2f(fmaf(s[r], sl2e, -L2)) : 0.f;
            s[r] = p * (dp[r] - Dq) * scale;  // dS
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const bf16x8_t dsf = pack_frag(s, st);
#pragma unroll
            for (int et = 0; et < 2; ++et) {
                const bf16x8_t ktf = lds_frag_tr(Kt + (et * 32 + l31) * VLD + kt * 32 + 16 * st + 4 * hi);
                dq[et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf, dq[et], 0, 0, 0);
            }
        }
    }
    if (q < P) store_rowT(dqkv + ((int64_t)f * P + q) * ld + h * D, dq, hi, accumulate);
}

__global__ __launch_bounds__(640) void attn_sp_bwd_dkv_mfma(const bf16* __restrict__ qkv, const bf16* __restrict__ o,
                                                             const bf16* __restrict__ d_o, const float* __restrict__ lse,
                                                             bf16* __restrict__ dqkv, int accumulate, int P, int H, float scale) {
    MAED_DYN_SHARED(unsigned short, smem);
    const int Pk = (P + 31) & ~31, VLD = Pk + 4;
    unsigned short* Qs = smem;
    unsigned short* dOs = Qs + (size_t)P * KLD;
    unsigned short* Qt = dOs + (size_t)P * KLD;
    unsigned short* dOt = Qt + (size_t)D * VLD;
    float* Ls = reinterpret_cast<float*>(dOt + (size_t)D * VLD);
    float* Ds = Ls + Pk;
    const int bid = attn_xcd_remap(blockIdx.x, gridDim.x);
    const int f = bid / H, h = bid % H, C = H * D;
    const int64_t ld = 3 * (int64_t)C;
    const bf16* base = qkv + (int64_t)f * P * ld + h * D;
    const bf16* obase = o + (int64_t)f * P * C + h * D;
    const bf16* dobase = d_o + (int64_t)f * P * C + h * D;
    const int tid = threadIdx.x, nthr = blockDim.x;
    stage_rowmajor_and_transposed(Qs, Qt, VLD, base, ld, P, tid, nthr);
    stage_rowmajor_and_transposed(dOs, dOt, VLD, dobase, C, P, tid, nthr);
    const float l2e = 1.44269504088896340736f;
    for (int qi = tid; qi < Pk; qi += nthr) {
        float dsum = 0.f, L = 0.f;
        if (qi < P) {
#pragma unroll
            for (int c = 0; c < D; c += 8) {
                float a[8], b[8];
                ld8(dobase + (int64_t)qi * C + c, a); ld8(obase + (int64_t)qi * C + c, b);
#pragma unroll
                for (int j = 0; j < 8; ++j) dsum = fmaf(a[j], b[j], dsum);
            }
            L = lse[((int64_t)f * H + h) * P + qi] * l2e;
        }
        Ds[qi] = dsum; Ls[qi] = L;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    if (wave * 32 >= P) return;
    const int k = wave * 32 + l31;
    const int kc = k < P ? k : P - 1;
    bf16x8_t kf[4], vf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        kf[t] = *reinterpret_cast<const bf16x8_t*>(base + C + (int64_t)kc * ld + t * 16 + hi * 8);
        vf[t] = *reinterpret_cast<const bf16x8_t*>(base + 2 * C + (int64_t)kc * ld + t * 16 + hi * 8);
    }
    const float sl2e = scale * l2e;
    f32x16_t dk[2], dv[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }
    const int nqt = Pk / 32;
    for (int qt = 0; qt < nqt; ++qt) {
        f32x16_t s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        int qrow = qt * 32 + l31;
        if (qrow > P - 1) qrow = P - 1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8_t qfr = *reinterpret_cast<const bf16x8_t*>(Qs + qrow * KLD + t * 16 + hi * 8);
            const bf16x8_t dofr = *reinterpret_cast<const bf16x8_t*>(dOs + qrow * KLD + t * 16 + hi * 8);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[t], s, 0, 0, 0);     // D[q][k]: lane = key
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dofr, vf[t], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const float p = (qq < P) ? __builtin_amdgcn_exp2f(fmaf(s[r], sl2e, -Ls[qq])) : 0.f;
            dp[r] = p * (dp[r] - Ds[qq]) * scale;  // dS
            s[r] = p;                              // P
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const bf16x8_t pf = pack_frag(s, st), dsf = pack_frag(dp, st);
#pragma unroll
            for (int et = 0; et < 2; ++et) {
                const int off = (et * 32 + l31) * VLD + qt * 32 + 16 * st + 4 * hi;
                dv[et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag_tr(dOt + off), pf, dv[et], 0, 0, 0);
                dk[et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag_tr(Qt + off), dsf, dk[et], 0, 0, 0);
            }
        }
    }
    if (k < P) {
        bf16* drow = dqkv + ((int64_t)f * P + k) * ld + h * D;
        store_rowT(drow + C, dk, hi, accumulate);
        store_rowT(drow + 2 * C, dv, hi, accumulate);
    }
}

// long-sequence (K/V-tiled) kernels: attn_long.hip
int maed_attn_long_fwd_launch(const void* qkv, void* o, float* lse, int F, int L, int H, float scale, hipStream_t s);
int maed_attn_long_bwd_launch(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int accumulate, int F, int L, int H,
                              float scale, hipStream_t s);
int maed_attn_long_fwd_valu_launch(const void* qkv, void* o, float* lse, int F, int L, int H, float scale, int dtype, hipStream_t s);
int maed_attn_long_bwd_valu_launch(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int accumulate, int F, int L,
                                   int H, float scale, int dtype, hipStream_t s);

// ==================================================================================================
static size_t valu_lds_bytes(int P, bool bwd_dkv) { return ((size_t)2 * P * LDF + (bwd_dkv ? 2 * P : 0)) * sizeof(float); }

extern "C" int maed_attn_spatial_fwd(const void* qkv, void* o, float* lse, int F, int P, int H, float scale, int dtype,
                                     int impl, void* stream) {
    MAED_CHECK_ARG(qkv && o && lse, MAED_ERR_ARG, "attn_spatial_fwd: null pointer");
    MAED_CHECK_ARG(F >= 0 && P > 0 && H > 0, MAED_ERR_SHAPE, "attn_spatial_fwd: bad extents");
    MAED_CHECK_ARG(is_aligned(qkv, 16) && is_aligned(o, 16), MAED_ERR_ALIGN, "attn_spatial_fwd: qkv/o must be 16-B aligned");
    if (F == 0) return MAED_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool use_mfma = (dtype == MAED_BF16) && (impl != MAED_IMPL_VALU);
    MAED_CHECK_ARG(!((impl == MAED_IMPL_MFMA || impl == MAED_IMPL_MFMA_LONG) && dtype != MAED_BF16), MAED_ERR_UNSUPPORTED,
                   "attn_spatial_fwd: the bf16 MFMA kernels need bf16 (f32: MAED_IMPL_X3 / _X6 or the process-wide fp32 matmul mode)");
    // fp32 q/k/v with split-bf16 contractions on the matrix cores (attn_x3.hip): explicitly, or when the process-wide fp32 matmul mode asks for it
    if (dtype == MAED_F32 && impl != MAED_IMPL_VALU) {
        const int np = impl == MAED_IMPL_X3 ? 2 : impl == MAED_IMPL_X6 ? 3 : maed_x3_planes();
        if (np) return maed_attn_x3_fwd_launch(np, qkv, o, lse, F, P, H, scale, s);
    }
    // the K/V-tiled kernels (128 query rows per workgroup, 64-key LDS tiles, lazy rescale): for sequences that do not fit one workgroup's
    // LDS (st_mode='coupling': T*P tokens) -- and by default, since they measured faster than the whole-head kernel at the frame sizes of
    // the path too (MI355X, profiles/r02_call1_attn_long_micro.txt: P = 197 33.9 vs 37.0 us, P = 257 71.7 vs 89.0 us).
    // MAED_IMPL_MFMA selects the whole-head kernel (A/B, tests).
    if (use_mfma && (impl == MAED_IMPL_MFMA_LONG || impl == MAED_IMPL_AUTO || P > 512)) return maed_attn_long_fwd_launch(qkv, o, lse, F, P, H, scale, s);
    if (use_mfma) {
        const int Pk = (P + 31) & ~31;
        MAED_CHECK_ARG(Pk / 32 <= 16, MAED_ERR_SHAPE, "attn_spatial_fwd(mfma): P=%d > 512 tokens per frame", P);
        const size_t lds = ((size_t)P * 64 + (size_t)D * (Pk + 4)) * 2;
        MAED_CHECK_ARG(lds <= 160 * 1024, MAED_ERR_SHAPE, "attn_spatial_fwd(mfma): LDS %zu B", lds);
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)attn_sp_fwd_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
        hipLaunchKernelGGL(attn_sp_fwd_mfma, dim3(F * H), dim3(64 * (Pk / 32)), lds, s, (const bf16*)qkv, (bf16*)o, lse, P, H,
                           scale * 1.44269504088896340736f);
    } else {
        const size_t lds = valu_lds_bytes(P, false);
        if (lds > 160 * 1024) return maed_attn_long_fwd_valu_launch(qkv, o, lse, F, P, H, scale, dtype, s);   // K/V do not fit: tiled
        if (dtype == MAED_F32) {
            (void)hipFuncSetAttribute((const void*)attn_sp_fwd_valu<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL((attn_sp_fwd_valu<float>), dim3(F * H), dim3(256), lds, s, (const float*)qkv, (float*)o, lse, P, H, scale);
        } else if (dtype == MAED_BF16) {
            (void)hipFuncSetAttribute((const void*)attn_sp_fwd_valu<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL((attn_sp_fwd_valu<bf16>), dim3(F * H), dim3(256), lds, s, (const bf16*)qkv, (bf16*)o, lse, P, H, scale);
        } else { maed_set_error("attn_spatial_fwd: bad dtype"); return MAED_ERR_ARG; }
    }
    MAED_CHECK_LAUNCH("attn_spatial_fwd");
    return MAED_OK;
}

template <typename T>
static void launch_bwd_valu(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int accumulate,
                            int F, int P, int H, float scale, hipStream_t s) {
    (void)hipFuncSetAttribute((const void*)attn_sp_bwd_dq_valu<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)attn_sp_bwd_dkv_valu<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((attn_sp_bwd_dq_valu<T>), dim3(F * H), dim3(256), valu_lds_bytes(P, false), s, (const T*)qkv, (const T*)o,
                       (const T*)d_o, lse, (T*)dqkv, accumulate, P, H, scale);
    hipLaunchKernelGGL((attn_sp_bwd_dkv_valu<T>), dim3(F * H), dim3(256), valu_lds_bytes(P, true), s, (const T*)qkv, (const T*)o,
                       (const T*)d_o, lse, (T*)dqkv, accumulate, P, H, scale);
}

extern "C" int maed_attn_spatial_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv,
                                     int accumulate, int F, int P, int H, float scale, int dtype, int impl, void* stream) {
    MAED_CHECK_ARG(qkv && o && d_o && lse && dqkv, MAED_ERR_ARG, "attn_spatial_bwd: null pointer");
    MAED_CHECK_ARG(F >= 0 && P > 0 && H > 0, MAED_ERR_SHAPE, "attn_spatial_bwd: bad extents");
    MAED_CHECK_ARG(!((impl == MAED_IMPL_MFMA || impl == MAED_IMPL_MFMA_LONG) && dtype != MAED_BF16), MAED_ERR_UNSUPPORTED,
                   "attn_spatial_bwd: the bf16 MFMA kernels need bf16 (f32: MAED_IMPL_X3 / _X6 or the process-wide fp32 matmul mode)");
    if (F == 0) return MAED_OK;
    if (dtype == MAED_F32 && impl != MAED_IMPL_VALU) {
        const int np = impl == MAED_IMPL_X3 ? 2 : impl == MAED_IMPL_X6 ? 3 : maed_x3_planes();
        if (np) return maed_attn_x3_bwd_launch(np, qkv, o, d_o, lse, dqkv, accumulate, F, P, H, scale, (hipStream_t)stream);
    }
    const bool mfma_fits = ((P + 31) / 32) <= 10;  // 640-thread workgroups (register budget 168/lane)
    // the tiled two-pass backward is the default for every length since its round-2 rework (delta from staged chunks, transposing LDS reads, trimmed
    // score arithmetic): P = 197 108.5 vs 147.7 us for the whole-head kernels, P = 257 (third row tile holds ONE row) 252.6 vs 314.1 us
    // (profiles/r02_attn_long_micro_v2.txt); the whole-head kernels stay reachable with impl = MAED_IMPL_MFMA
    // (round 4: a one-kernel backward per (frame, head) for P <= 224 measured 120 us against 106 for the tiled kernels: profiles/r04_attn_bwd_fused_rejected.txt)
    if (dtype == MAED_BF16 && impl != MAED_IMPL_VALU &&
        (impl == MAED_IMPL_MFMA_LONG || impl == MAED_IMPL_AUTO ||
         (!mfma_fits && (impl == MAED_IMPL_MFMA || valu_lds_bytes(P, true) > 160 * 1024))))
        return maed_attn_long_bwd_launch(qkv, o, d_o, lse, dqkv, accumulate, F, P, H, scale, (hipStream_t)stream);
    if (dtype == MAED_BF16 && impl != MAED_IMPL_VALU && mfma_fits) {
        const int Pk = (P + 31) & ~31;
        const size_t lds_dq = ((size_t)2 * P * KLD + (size_t)D * (Pk + 4)) * 2;
        const size_t lds_dkv = ((size_t)2 * P * KLD + (size_t)2 * D * (Pk + 4)) * 2 + (size_t)2 * Pk * sizeof(float);
        MAED_CHECK_ARG(lds_dkv <= 160 * 1024, MAED_ERR_SHAPE, "attn_spatial_bwd(mfma): P=%d needs %zu B LDS", P, lds_dkv);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)attn_sp_bwd_dq_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)attn_sp_bwd_dkv_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        hipStream_t s = (hipStream_t)stream;
        hipLaunchKernelGGL(attn_sp_bwd_dq_mfma, dim3(F * H), dim3(64 * (Pk / 32)), lds_dq, s, (const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse,
                           (bf16*)dqkv, accumulate, P, H, scale);
        hipLaunchKernelGGL(attn_sp_bwd_dkv_mfma, dim3(F * H), dim3(64 * (Pk / 32)), lds_dkv, s, (const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse,
                           (bf16*)dqkv, accumulate, P, H, scale);
        MAED_CHECK_LAUNCH("attn_spatial_bwd(mfma)");
        return MAED_OK;
    }
    if (valu_lds_bytes(P, true) > 160 * 1024)      // Q/dO do not fit one workgroup's LDS: tiled exact kernels
        return maed_attn_long_bwd_valu_launch(qkv, o, d_o, lse, dqkv, accumulate, F, P, H, scale, dtype, (hipStream_t)stream);
    if (dtype == MAED_F32) launch_bwd_valu<float>(qkv, o, d_o, lse, dqkv, accumulate, F, P, H, scale, (hipStream_t)stream);
    else if (dtype == MAED_BF16) launch_bwd_valu<bf16>(qkv, o, d_o, lse, dqkv, accumulate, F, P, H, scale, (hipStream_t)stream);
    else { maed_set_error("attn_spatial_bwd: bad dtype"); return MAED_ERR_ARG; }
    MAED_CHECK_LAUNCH("attn_spatial_bwd");
    return MAED_OK;
}
