// fp32-accurate attention on the bf16 matrix cores: the K/V-tiled flash forward and two-pass recompute backward of attn_long.hip with fp32 q/k/v/o in
// HBM and every contraction (S = Q K^T, O = P V; dP = dO V^T, dQ = dS K, dK = dS^T Q, dV = P^T dO) in split-bf16 arithmetic (gemm_x3.h: 3 MFMAs per
// product for "bf16x3", 6 for "bf16x6"), fp32 softmax statistics exactly as in the bf16 kernels.
//
// Reference: Attention.forward_spatial (lib/models/vision_transformer.py:206-214) in the reference's fp32 arithmetic; the f32 parity mode's
// exact VALU kernels (thread per row) take 543 us per cfg3 launch against 33 us for the bf16 MFMA kernel -- these keep fp32-level scores
// (the softmax exponent sees |error| ~2^-16 |q||k| / 8) at a few times the bf16 kernel's time.
//
// Same decomposition as attn_long.hip: workgroup = 128 rows (4 waves x 32) of one (item, head); the other side streamed in 64-row tiles.  A tile
// is loaded as fp32 (four 16-byte pieces per thread and tensor, prefetched into registers while the previous tile is consumed), split on the
// VALU and written as NP bf16 planes of the row-major LDS image (row stride KLD); transposed fragments come from ds_read_b64_tr_b16 on the same
// planes.  The lane-resident side (Q; Q and dO; K and V) is split once into register fragments; P / dS are split when they are packed.
#include "attn_mfma.cuh"
#include "gemm_x3.h"

#define D HEAD_DIM

namespace {

constexpr int LT = 64;              // streamed rows per LDS tile
constexpr int WG_ROWS = 128;        // rows of the owning side per workgroup
constexpr int PLANE = LT * KLD;     // elements of one bf16 plane of a tile

__device__ __forceinline__ int x3_xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// one streamed tile = rows [r0, r0 + 64) x 64 floats of src (row stride ld; rows past L-1 replicate row L-1: finite filler the callers mask):
// 1024 16-byte pieces, 4 per thread (piece idx = tid + 256 i: row idx >> 4, columns (idx & 15) * 4)
struct FTile { float4 a, b, c, d; };

__device__ __forceinline__ float4 ftile_load1(const float* src, int64_t ld, int r0, int L, int idx) {
    int row = r0 + (idx >> 4);
    if (row > L - 1) row = L - 1;
    return *reinterpret_cast<const float4*>(src + (int64_t)row * ld + (idx & 15) * 4);
}
__device__ __forceinline__ void ftile_load(FTile& t, const float* src, int64_t ld, int r0, int L, int tid) {
    t.a = ftile_load1(src, ld, r0, L, tid);
    t.b = ftile_load1(src, ld, r0, L, tid + 256);
    t.c = ftile_load1(src, ld, r0, L, tid + 512);
    t.d = ftile_load1(src, ld, r0, L, tid + 768);
}
template <int NP>
__device__ __forceinline__ void ftile_store1(const float4 v, unsigned short* planes, int idx) {
    uint2 pl[NP];
    split4<NP>(v.x, v.y, v.z, v.w, pl);
    unsigned short* dst = planes + (idx >> 4) * KLD + (idx & 15) * 4;
#pragma unroll
    for (int p = 0; p < NP; ++p) *reinterpret_cast<uint2*>(dst + p * PLANE) = pl[p];
}
template <int NP>
__device__ __forceinline__ void ftile_store(const FTile& t, unsigned short* planes, int tid) {
    ftile_store1<NP>(t.a, planes, tid);
    ftile_store1<NP>(t.b, planes, tid + 256);
    ftile_store1<NP>(t.c, planes, tid + 512);
    ftile_store1<NP>(t.d, planes, tid + 768);
}

// 8 consecutive fp32 of a lane's row -> the NP planes of one MFMA fragment
template <int NP>
__device__ __forceinline__ void load_split_frag(const float* p, bf16x8_t (&out)[NP]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    uint2 pa[NP], pb[NP];
    split4<NP>(a.x, a.y, a.z, a.w, pa);
    split4<NP>(b.x, b.y, b.z, b.w, pb);
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        union { bf16x8_t v; uint32_t u[4]; } f;
        f.u[0] = pa[q].x; f.u[1] = pa[q].y; f.u[2] = pb[q].x; f.u[3] = pb[q].y;
        out[q] = f.v;
    }
}
// accumulator values 8 st .. 8 st + 7 (the k-slot order pack_frag uses) -> NP fragment planes
template <int NP>
__device__ __forceinline__ void split_frag(const f32x16_t& x, int st, bf16x8_t (&out)[NP]) {
    uint2 pa[NP], pb[NP];
    split4<NP>(x[8 * st], x[8 * st + 1], x[8 * st + 2], x[8 * st + 3], pa);
    split4<NP>(x[8 * st + 4], x[8 * st + 5], x[8 * st + 6], x[8 * st + 7], pb);
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        union { bf16x8_t v; uint32_t u[4]; } f;
        f.u[0] = pa[q].x; f.u[1] = pa[q].y; f.u[2] = pb[q].x; f.u[3] = pb[q].y;
        out[q] = f.v;
    }
}
template <int NP>
__device__ __forceinline__ void lds_frag_planes(const unsigned short* p, bf16x8_t (&out)[NP]) {      // row-major fragment (8 consecutive k) of every plane
#pragma unroll
    for (int q = 0; q < NP; ++q) out[q] = *reinterpret_cast<const bf16x8_t*>(p + q * PLANE);
}
template <int NP>
__device__ __forceinline__ void lds_frag_tr_planes(const unsigned short* X, int key_base, int e_base, int lane, bf16x8_t (&out)[NP]) {   // transposed fragment
#pragma unroll
    for (int q = 0; q < NP; ++q) out[q] = lds_frag_tr_rm(X + q * PLANE, key_base, e_base, lane);
}
__device__ __forceinline__ void store_rowT_f32(float* row, const f32x16_t (&acc)[2], int hi, int accumulate) {
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int e0 = et * 32 + 8 * g + 4 * hi;
            float v[4] = {acc[et][4 * g], acc[et][4 * g + 1], acc[et][4 * g + 2], acc[et][4 * g + 3]};
            if (accumulate) { float o[4]; ld4(row + e0, o); v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3]; }
            st4(row + e0, v);
        }
}

template <int NP>
__global__ __launch_bounds__(256, 2) void attn_x3_fwd(const float* __restrict__ qkv, float* __restrict__ o, float* __restrict__ lse, int L, int H, int ntile,
                                                      float scale_log2e) {
    __shared__ __attribute__((aligned(16))) unsigned short Ks[NP * PLANE];
    __shared__ __attribute__((aligned(16))) unsigned short Vs[NP * PLANE];
    const int bid = x3_xcd_remap(blockIdx.x, gridDim.x);
    const int item = bid / ntile, tile = bid - item * ntile;
    const int f = item / H, h = item - f * H, C = H * D;
    const int64_t ld = 3 * (int64_t)C;
    const float* base = qkv + (int64_t)f * L * ld + h * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int q0 = tile * WG_ROWS + wave * 32, q = q0 + l31;
    const bool active = q0 < L;                     // wave-uniform; inactive waves still stage tiles and meet the barriers
    const int qc = q < L ? q : L - 1;
    bf16x8_t qf[4][NP];
#pragma unroll
    for (int t = 0; t < 4; ++t) load_split_frag<NP>(base + (int64_t)qc * ld + t * 16 + hi * 8, qf[t]);
    f32x16_t oacc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; }
    float m = -INFINITY, l = 0.f;
    const int nkt = (L + LT - 1) / LT;
    FTile kreg, vreg;
    ftile_load(kreg, base + C, ld, 0, L, tid);
    ftile_load(vreg, base + 2 * C, ld, 0, L, tid);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();                            // every wave is done with the previous tile
        ftile_store<NP>(kreg, Ks, tid);
        ftile_store<NP>(vreg, Vs, tid);
        __syncthreads();
        if (kt + 1 < nkt) {                         // next tile's loads fly while this one is consumed
            ftile_load(kreg, base + C, ld, (kt + 1) * LT, L, tid);
            ftile_load(vreg, base + 2 * C, ld, (kt + 1) * LT, L, tid);
        }
        if (!active) continue;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int k0 = kt * LT + sub * 32;
            if (k0 >= L) break;
            f32x16_t s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const unsigned short* kp = Ks + (sub * 32 + l31) * KLD + hi * 8;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bf16x8_t kf[NP];
                lds_frag_planes<NP>(kp + t * 16, kf);
                s = mfma_split<NP>(kf, qf[t], s);
            }
            if (k0 + 32 > L) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = (k0 + (r & 3) + 8 * (r >> 2) + 4 * hi < L) ? s[r] : -INFINITY;
            }
            float mt = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * scale_log2e;
            // lazy rescaling as in attn_long_fwd_mfma: the reference maximum moves only when it grew by more than 2^8
            const bool grow = mt > m + 8.0f;
            if (__any(grow)) {
                const float mn = grow ? mt : m;
                const float alpha = __builtin_amdgcn_exp2f(m - mn);
                l *= alpha;
                m = mn;
#pragma unroll
                for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }
            }
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, -m)); ps += s[r]; }
            l += ps;
#pragma unroll
            for (int st = 0; st < 2; ++st) {        // O^T += V^T P^T, 16 keys per step
                bf16x8_t pf[NP];
                split_frag<NP>(s, st, pf);
#pragma unroll
                for (int et = 0; et < 2; ++et) {
                    bf16x8_t vf[NP];
                    lds_frag_tr_planes<NP>(Vs, sub * 32 + 16 * st, et * 32, lane, vf);
                    oacc[et] = mfma_split<NP>(vf, pf, oacc[et]);
                }
            }
        }
    }
    l += __shfl_xor(l, 32, 64);
    if (active && q < L) {
        const float inv = 1.f / l;
        float* orow = o + ((int64_t)f * L + q) * C + h * D;
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(orow + et * 32 + 8 * g + 4 * hi) =
                    make_float4(oacc[et][4 * g] * inv, oacc[et][4 * g + 1] * inv, oacc[et][4 * g + 2] * inv, oacc[et][4 * g + 3] * inv);
        if (hi == 0) lse[((int64_t)f * H + h) * L + q] = (m + log2f(l)) * 0.69314718055994530942f;
    }
}

template <int NP>
__global__ __launch_bounds__(256, 2) void attn_x3_bwd_dq(const float* __restrict__ qkv, const float* __restrict__ o, const float* __restrict__ d_o,
                                                         const float* __restrict__ lse, float* __restrict__ dqkv, int accumulate, int L, int H, int ntile,
                                                         float scale) {
    __shared__ __attribute__((aligned(16))) unsigned short Ks[NP * PLANE];
    __shared__ __attribute__((aligned(16))) unsigned short Vs[NP * PLANE];
    const int bid = x3_xcd_remap(blockIdx.x, gridDim.x);
    const int item = bid / ntile, tile = bid - item * ntile;
    const int f = item / H, h = item - f * H, C = H * D;
    const int64_t ld = 3 * (int64_t)C;
    const float* base = qkv + (int64_t)f * L * ld + h * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int q0 = tile * WG_ROWS + wave * 32, q = q0 + l31;
    const bool active = q0 < L;
    const int qc = q < L ? q : L - 1;
    const float* orow = o + ((int64_t)f * L + qc) * C + h * D;
    const float* dorow = d_o + ((int64_t)f * L + qc) * C + h * D;
    bf16x8_t qf[4][NP], dof[4][NP];
    float Dq = 0.f;                                 // delta = rowsum(dO * O) of this lane's query (exact fp32)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        load_split_frag<NP>(base + (int64_t)qc * ld + t * 16 + hi * 8, qf[t]);
        load_split_frag<NP>(dorow + t * 16 + hi * 8, dof[t]);
        float a[8], b[8];
        ld8(dorow + t * 16 + hi * 8, a); ld8(orow + t * 16 + hi * 8, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) Dq = fmaf(a[j], b[j], Dq);
    }
    Dq += __shfl_xor(Dq, 32, 64);
    const float l2e = 1.44269504088896340736f;
    const float L2 = lse[((int64_t)f * H + h) * L + qc] * l2e, sl2e = scale * l2e;
    f32x16_t dq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }
    const int nkt = (L + LT - 1) / LT;
    FTile kreg, vreg;
    ftile_load(kreg, base + C, ld, 0, L, tid);
    ftile_load(vreg, base + 2 * C, ld, 0, L, tid);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        ftile_store<NP>(kreg, Ks, tid);
        ftile_store<NP>(vreg, Vs, tid);
        __syncthreads();
        if (kt + 1 < nkt) {
            ftile_load(kreg, base + C, ld, (kt + 1) * LT, L, tid);
            ftile_load(vreg, base + 2 * C, ld, (kt + 1) * LT, L, tid);
        }
        if (!active) continue;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int k0 = kt * LT + sub * 32;
            if (k0 >= L) break;
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = -Dq; }
            const int off = (sub * 32 + l31) * KLD + hi * 8;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bf16x8_t kf[NP], vf[NP];
                lds_frag_planes<NP>(Ks + off + t * 16, kf);
                lds_frag_planes<NP>(Vs + off + t * 16, vf);
                s = mfma_split<NP>(kf, qf[t], s);
                dp = mfma_split<NP>(vf, dof[t], dp);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], sl2e, -L2)) * dp[r];    // dS^T / scale
            if (k0 + 32 > L) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = (k0 + (r & 3) + 8 * (r >> 2) + 4 * hi < L) ? s[r] : 0.f;
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {        // dQ^T += K^T dS^T
                bf16x8_t dsf[NP];
                split_frag<NP>(s, st, dsf);
#pragma unroll
                for (int et = 0; et < 2; ++et) {
                    bf16x8_t ktf[NP];
                    lds_frag_tr_planes<NP>(Ks, sub * 32 + 16 * st, et * 32, lane, ktf);
                    dq[et] = mfma_split<NP>(ktf, dsf, dq[et]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] *= scale; dq[1][r] *= scale; }
    if (active && q < L) store_rowT_f32(dqkv + ((int64_t)f * L + q) * ld + h * D, dq, hi, accumulate);
}

template <int NP>
__global__ __launch_bounds__(256, 2) void attn_x3_bwd_dkv(const float* __restrict__ qkv, const float* __restrict__ o, const float* __restrict__ d_o,
                                                          const float* __restrict__ lse, float* __restrict__ dqkv, int accumulate, int L, int H, int ntile,
                                                          float scale) {
    __shared__ __attribute__((aligned(16))) unsigned short Qs[NP * PLANE];
    __shared__ __attribute__((aligned(16))) unsigned short dOs[NP * PLANE];
    __shared__ float Ls[LT], Ds[LT];
    const int bid = x3_xcd_remap(blockIdx.x, gridDim.x);
    const int item = bid / ntile, tile = bid - item * ntile;
    const int f = item / H, h = item - f * H, C = H * D;
    const int64_t ld = 3 * (int64_t)C;
    const float* base = qkv + (int64_t)f * L * ld + h * D;
    const float* obase = o + (int64_t)f * L * C + h * D;
    const float* dobase = d_o + (int64_t)f * L * C + h * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int k0w = tile * WG_ROWS + wave * 32, k = k0w + l31;
    const bool active = k0w < L;
    const int kc = k < L ? k : L - 1;
    bf16x8_t kf[4][NP], vf[4][NP];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        load_split_frag<NP>(base + C + (int64_t)kc * ld + t * 16 + hi * 8, kf[t]);
        load_split_frag<NP>(base + 2 * C + (int64_t)kc * ld + t * 16 + hi * 8, vf[t]);
    }
    const float l2e = 1.44269504088896340736f, sl2e = scale * l2e;
    f32x16_t dk[2], dv[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }
    const int nqt = (L + LT - 1) / LT;
    // delta = rowsum(dO * O) of the tile's queries from the dO pieces a thread stages anyway and the matching O pieces (16 threads share a row: four
    // shuffles); exact fp32
    FTile qreg, doreg;
    float lreg = 0.f;
    auto piece_dot = [&](const float4& a, int qt_, int idx) {
        const float4 b = ftile_load1(obase, C, qt_ * LT, L, idx);
        return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
    };
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
    auto load_all = [&](int qt_) {
        ftile_load(qreg, base, ld, qt_ * LT, L, tid);
        ftile_load(doreg, dobase, C, qt_ * LT, L, tid);
        d0 = piece_dot(doreg.a, qt_, tid); d1 = piece_dot(doreg.b, qt_, tid + 256); d2 = piece_dot(doreg.c, qt_, tid + 512); d3 = piece_dot(doreg.d, qt_, tid + 768);
        if (tid < LT) { int qi = qt_ * LT + tid; if (qi > L - 1) qi = L - 1; lreg = lse[((int64_t)f * H + h) * L + qi] * l2e; }
    };
    load_all(0);
    for (int qt = 0; qt < nqt; ++qt) {
        __syncthreads();
        ftile_store<NP>(qreg, Qs, tid);
        ftile_store<NP>(doreg, dOs, tid);
        {   // piece idx = tid + 256 i: row (tid >> 4) + 16 i; the 16 threads of a row are consecutive lanes
#pragma unroll
            for (int msk = 1; msk < 16; msk <<= 1) {
                d0 += __shfl_xor(d0, msk, 64); d1 += __shfl_xor(d1, msk, 64); d2 += __shfl_xor(d2, msk, 64); d3 += __shfl_xor(d3, msk, 64);
            }
            if ((tid & 15) == 0) { const int r = tid >> 4; Ds[r] = d0; Ds[16 + r] = d1; Ds[32 + r] = d2; Ds[48 + r] = d3; }
            if (tid < LT) Ls[tid] = lreg;
        }
        __syncthreads();
        if (qt + 1 < nqt) load_all(qt + 1);
        if (!active) continue;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int q0 = qt * LT + sub * 32;
            if (q0 >= L) break;
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = -Ds[sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi]; }
            const int off = (sub * 32 + l31) * KLD + hi * 8;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bf16x8_t qfr[NP], dofr[NP];
                lds_frag_planes<NP>(Qs + off + t * 16, qfr);
                lds_frag_planes<NP>(dOs + off + t * 16, dofr);
                s = mfma_split<NP>(qfr, kf[t], s);          // D[q][k]: lane = key
                dp = mfma_split<NP>(dofr, vf[t], dp);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], sl2e, -Ls[ql]));         // P
                dp[r] *= s[r];                                                     // dS / scale
            }
            if (q0 + 32 > L) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = q0 + (r & 3) + 8 * (r >> 2) + 4 * hi < L;
                    s[r] = ok ? s[r] : 0.f; dp[r] = ok ? dp[r] : 0.f;
                }
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8_t pf[NP], dsf[NP];
                split_frag<NP>(s, st, pf);
                split_frag<NP>(dp, st, dsf);
#pragma unroll
                for (int et = 0; et < 2; ++et) {
                    bf16x8_t dotf[NP], qtf[NP];
                    lds_frag_tr_planes<NP>(dOs, sub * 32 + 16 * st, et * 32, lane, dotf);
                    lds_frag_tr_planes<NP>(Qs, sub * 32 + 16 * st, et * 32, lane, qtf);
                    dv[et] = mfma_split<NP>(dotf, pf, dv[et]);       // dV^T += dO^T P
                    dk[et] = mfma_split<NP>(qtf, dsf, dk[et]);       // dK^T += Q^T dS
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[0][r] *= scale; dk[1][r] *= scale; }
    if (active && k < L) {
        float* drow = dqkv + ((int64_t)f * L + k) * ld + h * D;
        store_rowT_f32(drow + C, dk, hi, accumulate);
        store_rowT_f32(drow + 2 * C, dv, hi, accumulate);
    }
}


// ==================================================================================================
// Temporal attention (Attention.forward_temporal, vision_transformer.py:216-228) in the fp32-accurate mode: the one-tile kernels of attn_temporal.hip
// (attn_tm_fwd_mfma / attn_tm_bwd_mfma_l32) on fp32 operands with split-bf16 contractions.  A one-wave workgroup owns one (clip n, head h, group of
// G = 32 / T tokens): its 32 rows r = g*T + t are the T frames of G tokens (cfg3: T = 16, two tokens per tile); row r lives at frame n*T + t, token
// tg*G + g of the (F, P, 3C) qkv tensor.  With a single tile the "other side" of every product is the wave's own 32 rows: row-major operands (Q, K, V,
// dO fragments) are loaded straight from global memory as fp32, split once into NP register planes; only the A operands of the products that
// contract over rows (V^T; K^T, Q^T, dO^T) go through LDS, as row-major plane images read back transposed by ds_read_b64_tr_b16.  Keys of another
// token are masked (block-diagonal attention over the virtual sequence).  The f32 VALU kernels these replace in the split modes: 129 / 283 us per cfg3
// launch (forward / backward).
// ==================================================================================================
constexpr int TPLANE = 32 * KLD;    // one bf16 plane of a 32-row image

template <int NP>
__device__ __forceinline__ void tm_frag_tr_planes(const unsigned short* X, int key_base, int e_base, int lane, bf16x8_t (&out)[NP]) {
#pragma unroll
    for (int q = 0; q < NP; ++q) out[q] = lds_frag_tr_rm(X + q * TPLANE, key_base, e_base, lane);
}
// 8 consecutive fp32 of the lane's row -> NP register fragment planes (+ optionally the same planes into a row-major LDS image at dst)
template <int NP, bool TO_LDS>
__device__ __forceinline__ void tm_load_split(const float* p, bool ok, bf16x8_t (&out)[NP], unsigned short* dst) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (ok) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
    uint2 pa[NP], pb[NP];
    split4<NP>(a.x, a.y, a.z, a.w, pa);
    split4<NP>(b.x, b.y, b.z, b.w, pb);
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        union { bf16x8_t v; uint4 u; } f;
        f.u = make_uint4(pa[q].x, pa[q].y, pb[q].x, pb[q].y);
        out[q] = f.v;
        if constexpr (TO_LDS) *reinterpret_cast<uint4*>(dst + q * TPLANE) = f.u;
    }
}

#define TMX_ROW_OK(r) (p0 + (r) / Tn < P)
#define TMX_TOK(r) (((int64_t)n * Tn + (r) % Tn) * P + p0 + (r) / Tn)

template <int NP>
__global__ __launch_bounds__(64, 4) void attn_tm_x3_fwd(const float* __restrict__ qkv, float* __restrict__ o, float* __restrict__ lse, int P, int H, int Tn,
                                                        int G, int ngroups, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) unsigned short Vs[NP * TPLANE];
    const int C = H * D;
    const int64_t ld = 3 * (int64_t)C;
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    int bid = blockIdx.x;
    const int tg = bid % ngroups; bid /= ngroups;
    const int h = bid % H, n = bid / H;
    const int p0 = tg * G;
    const int row = l31;
    const bool row_ok = TMX_ROW_OK(row);
    const int64_t tok = row_ok ? TMX_TOK(row) : 0;
    const float* rp = qkv + tok * ld + h * D;
    bf16x8_t qf[4][NP], kf[4][NP];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int e0 = t * 16 + hi * 8;
        bf16x8_t vtmp[NP];
        tm_load_split<NP, false>(rp + e0, row_ok, qf[t], nullptr);
        tm_load_split<NP, false>(rp + C + e0, row_ok, kf[t], nullptr);
        tm_load_split<NP, true>(rp + 2 * C + e0, row_ok, vtmp, Vs + row * KLD + e0);     // rows of absent tokens are staged as zeros
    }
    __syncthreads();
    f32x16_t s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) s = mfma_split<NP>(kf[t], qf[t], s);                     // S^T: lane = query, registers = keys
    const int rg = row / Tn;
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = (r & 3) + 8 * (r >> 2) + 4 * hi;
        const bool ok = TMX_ROW_OK(k) && (k / Tn == rg);
        s[r] = ok ? s[r] * scale_log2e : -INFINITY;
        m = fmaxf(m, s[r]);
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float msafe = (m == -INFINITY) ? 0.f : m;                                      // (a query row of an absent token: everything masked)
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - msafe); l += s[r]; }
    l += __shfl_xor(l, 32, 64);
    f32x16_t oacc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; }
#pragma unroll
    for (int st = 0; st < 2; ++st) {                                                     // O^T = V^T P^T, 16 keys per step
        bf16x8_t pf[NP];
        split_frag<NP>(s, st, pf);
#pragma unroll
        for (int et = 0; et < 2; ++et) {
            bf16x8_t vf[NP];
            tm_frag_tr_planes<NP>(Vs, 16 * st, et * 32, lane, vf);
            oacc[et] = mfma_split<NP>(vf, pf, oacc[et]);
        }
    }
    if (row_ok) {
        const float inv = 1.f / l;
        float* orow = o + tok * C + h * D;
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(orow + et * 32 + 8 * g + 4 * hi) =
                    make_float4(oacc[et][4 * g] * inv, oacc[et][4 * g + 1] * inv, oacc[et][4 * g + 2] * inv, oacc[et][4 * g + 3] * inv);
        if (hi == 0) lse[(((int64_t)n * Tn + row % Tn) * H + h) * P + p0 + rg] = (m + log2f(l)) * 0.69314718055994530942f;
    }
}

template <int NP>
__global__ __launch_bounds__(64, 2) void attn_tm_x3_bwd(const float* __restrict__ qkv, const float* __restrict__ o, const float* __restrict__ d_o,
                                                        const float* __restrict__ lse, float* __restrict__ dqkv, int accumulate, int P, int H, int Tn, int G,
                                                        int ngroups, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned short Qs[NP * TPLANE], Ks[NP * TPLANE], dOs[NP * TPLANE];
    __shared__ float Ls[32], Ds[32];
    const int C = H * D;
    const int64_t ld = 3 * (int64_t)C;
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    int bid = blockIdx.x;
    const int tg = bid % ngroups; bid /= ngroups;
    const int h = bid % H, n = bid / H;
    const int p0 = tg * G;
    const int row = l31;
    const bool row_ok = TMX_ROW_OK(row);
    const int64_t tok = row_ok ? TMX_TOK(row) : 0;
    const float* rp = qkv + tok * ld + h * D;
    const float* gp = d_o + tok * C + h * D;
    const float* op = o + tok * C + h * D;
    const float l2e = 1.44269504088896340736f;
    bf16x8_t qf[4][NP], kf[4][NP], vf[4][NP], dof[4][NP];
    float dsum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int e0 = t * 16 + hi * 8;
        tm_load_split<NP, true>(rp + e0, row_ok, qf[t], Qs + row * KLD + e0);            // (rows of absent tokens: zeros)
        tm_load_split<NP, true>(rp + C + e0, row_ok, kf[t], Ks + row * KLD + e0);
        tm_load_split<NP, false>(rp + 2 * C + e0, row_ok, vf[t], nullptr);
        tm_load_split<NP, true>(gp + e0, row_ok, dof[t], dOs + row * KLD + e0);
        if (row_ok) {
            float a[8], b[8];
            ld8(gp + e0, a); ld8(op + e0, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) dsum = fmaf(a[j], b[j], dsum);
        }
    }
    dsum += __shfl_xor(dsum, 32, 64);
    const float Dq = dsum;
    const int rg = row / Tn;
    const float L2 = row_ok ? lse[(((int64_t)n * Tn + row % Tn) * H + h) * P + p0 + rg] * l2e : 0.f;
    if (hi == 0) { Ds[row] = Dq; Ls[row] = L2; }
    __syncthreads();

    const float sl2e = scale * l2e;
    float* drow = dqkv + tok * ld + h * D;
    {   // ---- pass A: lane = query;  S^T = K Q^T, dP^T = V dO^T ----
        f32x16_t sa, dp, dq[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = -Dq; dq[0][r] = 0.f; dq[1][r] = 0.f; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            sa = mfma_split<NP>(kf[t], qf[t], sa);
            dp = mfma_split<NP>(vf[t], dof[t], dp);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const bool ok = row_ok && TMX_ROW_OK(k) && (k / Tn == rg);
            const float pr = ok ? __builtin_amdgcn_exp2f(fmaf(sa[r], sl2e, -L2)) : 0.f;
            sa[r] = pr * dp[r];                      // dS / scale
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            bf16x8_t dsf[NP];
            split_frag<NP>(sa, st, dsf);
#pragma unroll
            for (int et = 0; et < 2; ++et) {
                bf16x8_t ktf[NP];
                tm_frag_tr_planes<NP>(Ks, 16 * st, et * 32, lane, ktf);
                dq[et] = mfma_split<NP>(ktf, dsf, dq[et]);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { dq[0][r] *= scale; dq[1][r] *= scale; }
        if (row_ok) store_rowT_f32(drow, dq, hi, accumulate);
    }
    {   // ---- pass B: lane = key;  S = Q K^T, dP = dO V^T ----
        f32x16_t sb, dp, dk[2], dv[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { sb[r] = 0.f; dp[r] = -Ds[(r & 3) + 8 * (r >> 2) + 4 * hi]; dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            sb = mfma_split<NP>(qf[t], kf[t], sb);
            dp = mfma_split<NP>(dof[t], vf[t], dp);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const bool ok = row_ok && TMX_ROW_OK(qq) && (qq / Tn == rg);
            const float pr = ok ? __builtin_amdgcn_exp2f(fmaf(sb[r], sl2e, -Ls[qq])) : 0.f;
            dp[r] *= pr;                             // dS / scale
            sb[r] = pr;                              // P
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            bf16x8_t pf[NP], dsf[NP];
            split_frag<NP>(sb, st, pf);
            split_frag<NP>(dp, st, dsf);
#pragma unroll
            for (int et = 0; et < 2; ++et) {
                bf16x8_t dotf[NP], qtf[NP];
                tm_frag_tr_planes<NP>(dOs, 16 * st, et * 32, lane, dotf);
                tm_frag_tr_planes<NP>(Qs, 16 * st, et * 32, lane, qtf);
                dv[et] = mfma_split<NP>(dotf, pf, dv[et]);       // dV^T += dO^T P
                dk[et] = mfma_split<NP>(qtf, dsf, dk[et]);       // dK^T += Q^T dS
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[0][r] *= scale; dk[1][r] *= scale; }
        if (row_ok) {
            store_rowT_f32(drow + C, dk, hi, accumulate);
            store_rowT_f32(drow + 2 * C, dv, hi, accumulate);
        }
    }
}
#undef TMX_ROW_OK
#undef TMX_TOK

}  // namespace

int maed_attn_x3_fwd_launch(int np, const void* qkv, void* o, float* lse, int F, int L, int H, float scale, hipStream_t s) {
    const int ntile = (L + WG_ROWS - 1) / WG_ROWS;
    MAED_CHECK_ARG((int64_t)F * H * ntile < (1ll << 31), MAED_ERR_SHAPE, "attn_x3_fwd: grid too large");
    const dim3 grid((unsigned)(F * H * ntile));
    const float sl2e = scale * 1.44269504088896340736f;
    if (np == 3) hipLaunchKernelGGL(attn_x3_fwd<3>, grid, dim3(256), 0, s, (const float*)qkv, (float*)o, lse, L, H, ntile, sl2e);
    else hipLaunchKernelGGL(attn_x3_fwd<2>, grid, dim3(256), 0, s, (const float*)qkv, (float*)o, lse, L, H, ntile, sl2e);
    MAED_CHECK_LAUNCH("attn_x3_fwd");
    return MAED_OK;
}

int maed_attn_x3_bwd_launch(int np, const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int accumulate, int F, int L, int H, float scale,
                            hipStream_t s) {
    const int ntile = (L + WG_ROWS - 1) / WG_ROWS;
    MAED_CHECK_ARG((int64_t)F * H * ntile < (1ll << 31), MAED_ERR_SHAPE, "attn_x3_bwd: grid too large");
    const dim3 grid((unsigned)(F * H * ntile));
#define X3_BWD(NP_) \
    hipLaunchKernelGGL(attn_x3_bwd_dq<NP_>, grid, dim3(256), 0, s, (const float*)qkv, (const float*)o, (const float*)d_o, lse, (float*)dqkv, accumulate, L, H, ntile, scale); \
    hipLaunchKernelGGL(attn_x3_bwd_dkv<NP_>, grid, dim3(256), 0, s, (const float*)qkv, (const float*)o, (const float*)d_o, lse, (float*)dqkv, accumulate, L, H, ntile, scale);
    if (np == 3) { X3_BWD(3) } else { X3_BWD(2) }
#undef X3_BWD
    MAED_CHECK_LAUNCH("attn_x3_bwd");
    return MAED_OK;
}

// temporal attention, one-tile virtual sequences (32 % T == 0: T = 16 packs two tokens per tile), bf16x3 only (two planes): false = not this shape / engine,
// the caller keeps its exact kernels (never less accurate than asked for)
bool maed_attn_tm_x3_fwd_launch(int np, const void* qkv, void* o, float* lse, int F, int P, int H, int T, float scale, hipStream_t s) {
    if (np != 2 || T > 32 || 32 % T != 0) return false;
    const int G = 32 / T, ngroups = (P + G - 1) / G;
    if ((int64_t)(F / T) * H * ngroups >= (1ll << 31)) return false;
    hipLaunchKernelGGL(attn_tm_x3_fwd<2>, dim3((unsigned)((F / T) * H * ngroups)), dim3(64), 0, s, (const float*)qkv, (float*)o, lse, P, H, T, G, ngroups,
                       scale * 1.44269504088896340736f);
    return true;
}

bool maed_attn_tm_x3_bwd_launch(int np, const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int accumulate, int F, int P, int H, int T,
                                float scale, hipStream_t s) {
    if (np != 2 || T > 32 || 32 % T != 0) return false;
    const int G = 32 / T, ngroups = (P + G - 1) / G;
    if ((int64_t)(F / T) * H * ngroups >= (1ll << 31)) return false;
    hipLaunchKernelGGL(attn_tm_x3_bwd<2>, dim3((unsigned)((F / T) * H * ngroups)), dim3(64), 0, s, (const float*)qkv, (const float*)o, (const float*)d_o, lse,
                       (float*)dqkv, accumulate, P, H, T, G, ngroups, scale);
    return true;
}
