// Long-sequence attention on MFMA (bf16): K/V-tiled flash forward and the two-pass recompute backward.
//
// Why: st_mode='coupling' (reference lib/models/vision_transformer.py:160-163,180-204) attends over the T*P tokens of a
// clip -- 16 x 197 = 3152 at cfg3 -- and the whole-head kernels of attn_spatial.hip keep one head's K/V (or Q/dO) resident in
// LDS, which stops at 512 (forward) / 320 (backward) tokens.  Same arithmetic, same operand orientation, same fragment tricks
// as those kernels (S^T = K Q^T so softmax statistics are per lane; P^T / dS packed straight from the accumulators as the
// next MFMA's B operand), but the resident side is streamed through LDS in 64-row tiles and a workgroup owns 128 rows
// (4 waves x 32) of the other side, so the grid is (sequence / 128) x (items x heads) workgroups of 256 threads at 18-36 KB
// of LDS each instead of items x heads workgroups of up to 1024 threads.
//
//   attn_long_fwd_mfma     wave = 32 queries; K tile row-major + V tile transposed in LDS; online softmax
//   attn_long_bwd_dq_mfma  wave = 32 queries; K (row-major + transposed) and V tiles in LDS
//   attn_long_bwd_dkv_mfma wave = 32 keys;    Q and dO tiles (row-major + transposed), lse and delta = rowsum(dO*O) in LDS
//
// Data layout is the spatial kernels': qkv (items, L, 3C) with q|k|v channel blocks and head-major channels, o / d_o (items, L, C),
// lse (items, H, L) natural log, dqkv like qkv.  Selected by maed_attn_spatial_{fwd,bwd} when the sequence does not fit the
// whole-head kernels, or explicitly with impl = MAED_IMPL_MFMA_LONG.  One LDS buffer, two barriers per tile; the next tile's global
// loads are issued into registers before the current tile is consumed (register prefetch), so only the LDS write sits between the
// barriers.  Measured on MI355X in round 2 (profiles/r02_attn_long_micro_v2.txt): P = 197 forward 30.4 us, backward 108 us; since then the default
// for every sequence length.
#include "attn_mfma.cuh"

#define D HEAD_DIM

namespace {

constexpr int LT = 64;          // rows of the streamed side per LDS tile (two 32-row MFMA sub-tiles)
constexpr int LVLD = LT + 4;    // row stride of transposed images: 34 dwords = 2 * odd -> conflict-free ds_read_b64
constexpr int WG_ROWS = 128;    // rows of the owning side per workgroup (4 waves x 32)

// consecutive logical workgroups (the row tiles of one (item, head), then the next head ...) on ONE XCD: they stream the same
// K/V (or Q/dO) tiles, which then stay in that XCD's L2
__device__ __forceinline__ int long_xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// One streamed tile = rows [r0, r0+64) of src (64 bf16 per row, row stride ld elements; rows past L-1 replicate row L-1: finite
// filler that the callers mask).  256 threads x 2 chunks of 16 B.  Split in two so that the NEXT tile's global loads are in
// flight while the current tile is consumed: tile_load() right after the barrier that publishes the current tile, tile_store()
// (row-major image with stride KLD and/or transposed image with stride LVLD) after the barrier that retires it.
// (two named uint4 per tile rather than an array: arrays that live across the loop's conditional reload get demoted to scratch)
struct TileRegs { uint4 a, b; };

__device__ __forceinline__ uint4 tile_load1(const bf16* src, int64_t ld, int r0, int L, int idx) {
    int row = r0 + (idx >> 3);
    if (row > L - 1) row = L - 1;
    return *reinterpret_cast<const uint4*>(src + (int64_t)row * ld + (idx & 7) * 8);
}
__device__ __forceinline__ void tile_load(TileRegs& t, const bf16* src, int64_t ld, int r0, int L, int tid) {
    t.a = tile_load1(src, ld, r0, L, tid);
    t.b = tile_load1(src, ld, r0, L, tid + 256);
}

__device__ __forceinline__ void tile_store1(const uint4 v, unsigned short* rows, unsigned short* tr, int idx) {
    const int p = idx >> 3, c8 = (idx & 7) * 8;
    if (rows) *reinterpret_cast<uint4*>(rows + p * KLD + c8) = v;
    if (tr) {
        tr[(c8 + 0) * LVLD + p] = (unsigned short)(v.x & 0xffffu); tr[(c8 + 1) * LVLD + p] = (unsigned short)(v.x >> 16);
        tr[(c8 + 2) * LVLD + p] = (unsigned short)(v.y & 0xffffu); tr[(c8 + 3) * LVLD + p] = (unsigned short)(v.y >> 16);
        tr[(c8 + 4) * LVLD + p] = (unsigned short)(v.z & 0xffffu); tr[(c8 + 5) * LVLD + p] = (unsigned short)(v.z >> 16);
        tr[(c8 + 6) * LVLD + p] = (unsigned short)(v.w & 0xffffu); tr[(c8 + 7) * LVLD + p] = (unsigned short)(v.w >> 16);
    }
}
__device__ __forceinline__ void tile_store(const TileRegs& t, unsigned short* rows, unsigned short* tr, int tid) {
    tile_store1(t.a, rows, tr, tid);
    tile_store1(t.b, rows, tr, tid + 256);
}

// (occupancy hints: without them the 512-register budget of a 256-thread workgroup makes the compiler park the MFMA accumulators in
// AGPRs and copy all 32 of them to VGPRs and back around every softmax rescale -- 159 v_accvgpr_read + 159 write in this kernel)
__global__ __launch_bounds__(256, 4) void attn_long_fwd_mfma(const bf16* __restrict__ qkv, bf16* __restrict__ o, float* __restrict__ lse,
                                                          int L, int H, int ntile, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) unsigned short Ks[LT * KLD];
    __shared__ __attribute__((aligned(16))) unsigned short Vs[LT * KLD];      // row-major: V^T fragments come from transposing reads
    const int bid = long_xcd_remap(blockIdx.x, gridDim.x);
    const int item = bid / ntile, tile = bid - item * ntile;
    const int f = item / H, h = item - f * H, C = H * D;
    const int64_t ld = 3 * (int64_t)C;
    const bf16* base = qkv + (int64_t)f * L * ld + h * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int q0 = tile * WG_ROWS + wave * 32, q = q0 + l31;
    const bool active = q0 < L;                     // wave-uniform; inactive waves still stage tiles and meet the barriers
    const int qc = q < L ? q : L - 1;
    bf16x8_t qf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) qf[t] = *reinterpret_cast<const bf16x8_t*>(base + (int64_t)qc * ld + t * 16 + hi * 8);
    f32x16_t oacc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; }
    float m = -INFINITY, l = 0.f;
    const int nkt = (L + LT - 1) / LT;
    TileRegs kreg, vreg;
    tile_load(kreg, base + C, ld, 0, L, tid);
    tile_load(vreg, base + 2 * C, ld, 0, L, tid);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();                            // every wave is done with the previous tile
        tile_store(kreg, Ks, nullptr, tid);
        tile_store(vreg, Vs, nullptr, tid);
        __syncthreads();
        if (kt + 1 < nkt) {                         // next tile's loads fly while this one is consumed
            tile_load(kreg, base + C, ld, (kt + 1) * LT, L, tid);
            tile_load(vreg, base + 2 * C, ld, (kt + 1) * LT, L, tid);
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
            for (int t = 0; t < 4; ++t)
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(kp + t * 16), qf[t], s, 0, 0, 0);
            // lane holds keys k0 + (r&3) + 8*(r>>2) + 4*hi of its query; only the sequence's last sub-tile has padding keys
            if (k0 + 32 > L) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = (k0 + (r & 3) + 8 * (r >> 2) + 4 * hi < L) ? s[r] : -INFINITY;
            }
            float mt = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * scale_log2e;
            // lazy rescaling: the softmax reference m of a query moves only when its running maximum grew by more than 2^8 -- the
            // probabilities are then at most 256 (exact in fp32 / bf16 all the same; o = acc / l uses the same reference), and after the
            // first tile or two no lane of the wave qualifies any more, so the 32 accumulator multiplies, the exp2 of the correction
            // and the bookkeeping are skipped wave-uniformly for most tiles (they are ~a fifth of the VALU work per tile)
            const bool grow = mt > m + 8.0f;
            if (__any(grow)) {
                const float mn = grow ? mt : m;
                const float alpha = __builtin_amdgcn_exp2f(m - mn);     // 1 for the lanes that stay
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
                const bf16x8_t pf = pack_frag(s, st);
#pragma unroll
                for (int et = 0; et < 2; ++et)
                    oacc[et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag_tr_rm(Vs, sub * 32 + 16 * st, et * 32, lane), pf, oacc[et], 0, 0, 0);
            }
        }
    }
    l += __shfl_xor(l, 32, 64);
    __syncthreads();                                // the K / V tiles are retired: their LDS carries the waves' output patches (full-line stores, attn_mfma.cuh)
    if (active) {
        const float inv = 1.f / l;
#pragma unroll
        for (int r = 0; r < 16; ++r) { oacc[0][r] *= inv; oacc[1][r] *= inv; }
        store_tile_lines((wave < 2 ? Ks : Vs) + (wave & 1) * 2048, o + ((int64_t)f * L + q0) * C + h * D, C, L - q0, oacc, lane, 0);
    }
    if (active && q < L) {
        if (hi == 0) lse[((int64_t)f * H + h) * L + q] = (m + log2f(l)) * 0.69314718055994530942f;
    }
}

__global__ __launch_bounds__(256, 3) void attn_long_bwd_dq_mfma(const bf16* __restrict__ qkv, const bf16* __restrict__ o, const bf16* __restrict__ d_o,
                                                             const float* __restrict__ lse, bf16* __restrict__ dqkv, int accumulate, int L, int H,
                                                             int ntile, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned short Ks[LT * KLD];
    __shared__ __attribute__((aligned(16))) unsigned short Vs[LT * KLD];
    const int bid = long_xcd_remap(blockIdx.x, gridDim.x);
    const int item = bid / ntile, tile = bid - item * ntile;
    const int f = item / H, h = item - f * H, C = H * D;
    const int64_t ld = 3 * (int64_t)C;
    const bf16* base = qkv + (int64_t)f * L * ld + h * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int q0 = tile * WG_ROWS + wave * 32, q = q0 + l31;
    const bool active = q0 < L;
    const int qc = q < L ? q : L - 1;
    const bf16* orow = o + ((int64_t)f * L + qc) * C + h * D;
    const bf16* dorow = d_o + ((int64_t)f * L + qc) * C + h * D;
    bf16x8_t qf[4], dof[4];
    float Dq = 0.f;                                 // delta = rowsum(dO * O) of this lane's query
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
    const float L2 = lse[((int64_t)f * H + h) * L + qc] * l2e, sl2e = scale * l2e;
    f32x16_t dq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }
    const int nkt = (L + LT - 1) / LT;
    TileRegs kreg, vreg;
    tile_load(kreg, base + C, ld, 0, L, tid);
    tile_load(vreg, base + 2 * C, ld, 0, L, tid);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        tile_store(kreg, Ks, nullptr, tid);
        tile_store(vreg, Vs, nullptr, tid);
        __syncthreads();
        if (kt + 1 < nkt) {
            tile_load(kreg, base + C, ld, (kt + 1) * LT, L, tid);
            tile_load(vreg, base + 2 * C, ld, (kt + 1) * LT, L, tid);
        }
        if (!active) continue;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int k0 = kt * LT + sub * 32;
            if (k0 >= L) break;
            // per score: one fma, one exp2, one multiply.  delta rides in the dP accumulator's initial value (a lane owns one query), the
            // 1/sqrt(d) factor is applied once to the dQ accumulators at the end, and only the sequence's last sub-tile masks padding keys
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = -Dq; }
            const int off = (sub * 32 + l31) * KLD + hi * 8;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(Ks + off + t * 16), qf[t], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(Vs + off + t * 16), dof[t], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], sl2e, -L2)) * dp[r];    // dS^T / scale
            if (k0 + 32 > L) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = (k0 + (r & 3) + 8 * (r >> 2) + 4 * hi < L) ? s[r] : 0.f;
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {        // dQ^T += K^T dS^T
                const bf16x8_t dsf = pack_frag(s, st);
#pragma unroll
                for (int et = 0; et < 2; ++et)
                    dq[et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag_tr_rm(Ks, sub * 32 + 16 * st, et * 32, lane), dsf, dq[et], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] *= scale; dq[1][r] *= scale; }
    __syncthreads();                                // (as in the forward: result tiles leave through the retired tile buffers)
    if (active) store_tile_lines((wave < 2 ? Ks : Vs) + (wave & 1) * 2048, dqkv + ((int64_t)f * L + q0) * ld + h * D, ld, L - q0, dq, lane, accumulate);
}

__global__ __launch_bounds__(256, 2) void attn_long_bwd_dkv_mfma(const bf16* __restrict__ qkv, const bf16* __restrict__ o, const bf16* __restrict__ d_o,
                                                              const float* __restrict__ lse, bf16* __restrict__ dqkv, int accumulate, int L, int H,
                                                              int ntile, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned short Qs[LT * KLD];
    __shared__ __attribute__((aligned(16))) unsigned short dOs[LT * KLD];
    __shared__ float Ls[LT], Ds[LT];
    const int bid = long_xcd_remap(blockIdx.x, gridDim.x);
    const int item = bid / ntile, tile = bid - item * ntile;
    const int f = item / H, h = item - f * H, C = H * D;
    const int64_t ld = 3 * (int64_t)C;
    const bf16* base = qkv + (int64_t)f * L * ld + h * D;
    const bf16* obase = o + (int64_t)f * L * C + h * D;
    const bf16* dobase = d_o + (int64_t)f * L * C + h * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int k0w = tile * WG_ROWS + wave * 32, k = k0w + l31;
    const bool active = k0w < L;
    const int kc = k < L ? k : L - 1;
    bf16x8_t kf[4], vf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        kf[t] = *reinterpret_cast<const bf16x8_t*>(base + C + (int64_t)kc * ld + t * 16 + hi * 8);
        vf[t] = *reinterpret_cast<const bf16x8_t*>(base + 2 * C + (int64_t)kc * ld + t * 16 + hi * 8);
    }
    const float l2e = 1.44269504088896340736f, sl2e = scale * l2e;
    f32x16_t dk[2], dv[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }
    const int nqt = (L + LT - 1) / LT;
    // delta = rowsum(dO * O) of the tile's queries comes from the dO chunks a thread stages anyway and the matching O chunks, prefetched with them
    // (8 threads share a row: three shuffles) -- the first version had 64 threads re-read dO and O between the two barriers of every tile, an
    // exposed global-load latency per tile
    TileRegs qreg, doreg, oreg;
    float lreg = 0.f;
    auto load_stats = [&](int qt_) {
        tile_load(oreg, obase, C, qt_ * LT, L, tid);
        if (tid < LT) { int qi = qt_ * LT + tid; if (qi > L - 1) qi = L - 1; lreg = lse[((int64_t)f * H + h) * L + qi] * l2e; }
    };
    auto chunk_dot = [](const uint4& a, const uint4& b) {
        const uint32_t x[4] = {a.x, a.y, a.z, a.w}, y[4] = {b.x, b.y, b.z, b.w};
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            d = fmaf(__uint_as_float(x[j] << 16), __uint_as_float(y[j] << 16), d);
            d = fmaf(__uint_as_float(x[j] & 0xffff0000u), __uint_as_float(y[j] & 0xffff0000u), d);
        }
        return d;
    };
    tile_load(qreg, base, ld, 0, L, tid);
    tile_load(doreg, dobase, C, 0, L, tid);
    load_stats(0);
    for (int qt = 0; qt < nqt; ++qt) {
        __syncthreads();
        tile_store(qreg, Qs, nullptr, tid);
        tile_store(doreg, dOs, nullptr, tid);
        {   // chunk idx = tid (row tid >> 3) and tid + 256 (row 32 + (tid >> 3)); the 8 threads of a row are consecutive lanes
            float d0 = chunk_dot(doreg.a, oreg.a), d1 = chunk_dot(doreg.b, oreg.b);
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) { d0 += __shfl_xor(d0, m, 64); d1 += __shfl_xor(d1, m, 64); }
            if ((tid & 7) == 0) { Ds[tid >> 3] = d0; Ds[32 + (tid >> 3)] = d1; }
            if (tid < LT) Ls[tid] = lreg;
        }
        __syncthreads();
        if (qt + 1 < nqt) {
            tile_load(qreg, base, ld, (qt + 1) * LT, L, tid);
            tile_load(doreg, dobase, C, (qt + 1) * LT, L, tid);
            load_stats(qt + 1);
        }
        if (!active) continue;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int q0 = qt * LT + sub * 32;
            if (q0 >= L) break;
            // as in the dQ pass: -delta is the dP accumulator's initial value (here a register owns a query: read from LDS), the 1/sqrt(d) factor is
            // applied once to the dK accumulators at the end, padding queries are masked in the sequence's last sub-tile only
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = -Ds[sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi]; }
            const int off = (sub * 32 + l31) * KLD + hi * 8;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(Qs + off + t * 16), kf[t], s, 0, 0, 0);      // D[q][k]: lane = key
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(dOs + off + t * 16), vf[t], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;         // row within the tile
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
                const bf16x8_t pf = pack_frag(s, st), dsf = pack_frag(dp, st);
#pragma unroll
                for (int et = 0; et < 2; ++et) {
                    dv[et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag_tr_rm(dOs, sub * 32 + 16 * st, et * 32, lane), pf, dv[et], 0, 0, 0);    // dV^T += dO^T P
                    dk[et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag_tr_rm(Qs, sub * 32 + 16 * st, et * 32, lane), dsf, dk[et], 0, 0, 0);     // dK^T += Q^T dS
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[0][r] *= scale; dk[1][r] *= scale; }
    __syncthreads();
    if (active) {
        unsigned short* patch = (wave < 2 ? Qs : dOs) + (wave & 1) * 2048;
        bf16* drow0 = dqkv + ((int64_t)f * L + k0w) * ld + h * D;
        store_tile_lines(patch, drow0 + C, ld, L - k0w, dk, lane, accumulate);
        store_tile_lines(patch, drow0 + 2 * C, ld, L - k0w, dv, lane, accumulate);
    }
}

// ==================================================================================================
// Exact-arithmetic (VALU) variants for the f32 parity mode: thread per row, the streamed side in fp32 LDS tiles.
// Same decomposition as the whole-head VALU kernels of attn_spatial.hip; correctness path, not a throughput path.
// ==================================================================================================
constexpr int VT = 64;          // streamed rows per tile
constexpr int VLDF = 65;        // padded fp32 LDS row
constexpr int VROWS = 256;      // rows owned by a workgroup (one per thread)

template <typename T>
__device__ __forceinline__ void stage_tile_f32(float* dst, const T* src, int64_t ld, int r0, int L, int tid) {
    for (int idx = tid; idx < VT * (D / 4); idx += 256) {
        const int p = idx / (D / 4), c = (idx % (D / 4)) * 4;
        int row = r0 + p;
        if (row > L - 1) row = L - 1;
        float v[4];
        ld4(src + (int64_t)row * ld + c, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[p * VLDF + c + j] = v[j];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_long_fwd_valu(const T* __restrict__ qkv, T* __restrict__ o, float* __restrict__ lse, int L, int H,
                                                          int ntile, float scale) {
    __shared__ float Ks[VT * VLDF], Vs[VT * VLDF];
    const int item = blockIdx.x / ntile, tile = blockIdx.x - item * ntile;
    const int f = item / H, h = item - f * H, C = H * D;
    const int64_t ld = 3 * (int64_t)C;
    const T* base = qkv + (int64_t)f * L * ld + h * D;
    const int tid = threadIdx.x, q = tile * VROWS + tid;
    const int qc = q < L ? q : L - 1;
    float qv[D], acc[D];
#pragma unroll
    for (int c = 0; c < D; c += 4) { float v[4]; ld4(base + (int64_t)qc * ld + c, v); qv[c] = v[0]; qv[c + 1] = v[1]; qv[c + 2] = v[2]; qv[c + 3] = v[3]; }
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = 0.f;
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < L; k0 += VT) {
        __syncthreads();
        stage_tile_f32(Ks, base + C, ld, k0, L, tid);
        stage_tile_f32(Vs, base + 2 * C, ld, k0, L, tid);
        __syncthreads();
        const int kn = (L - k0 < VT) ? L - k0 : VT;
        for (int k = 0; k < kn; ++k) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) s = fmaf(qv[c], Ks[k * VLDF + c], s);
            s *= scale;
            const float mn = fmaxf(m, s);
            const float a = __expf(m - mn), p = __expf(s - mn);
            l = l * a + p;
#pragma unroll
            for (int c = 0; c < D; ++c) acc[c] = fmaf(p, Vs[k * VLDF + c], acc[c] * a);
            m = mn;
        }
    }
    if (q < L) {
        const float inv = 1.f / l;
        T* orow = o + ((int64_t)f * L + q) * C + h * D;
#pragma unroll
        for (int c = 0; c < D; c += 4) { float v[4] = {acc[c] * inv, acc[c + 1] * inv, acc[c + 2] * inv, acc[c + 3] * inv}; st4(orow + c, v); }
        lse[((int64_t)f * H + h) * L + q] = m + __logf(l);
    }
}

template <typename T>
__device__ __forceinline__ void store_row_acc(T* dst, const float (&acc)[D], int accumulate) {
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        float v[4] = {acc[c], acc[c + 1], acc[c + 2], acc[c + 3]};
        if (accumulate) { float old[4]; ld4(dst + c, old); v[0] += old[0]; v[1] += old[1]; v[2] += old[2]; v[3] += old[3]; }
        st4(dst + c, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_long_bwd_dq_valu(const T* __restrict__ qkv, const T* __restrict__ o, const T* __restrict__ d_o,
                                                             const float* __restrict__ lse, T* __restrict__ dqkv, int accumulate, int L, int H,
                                                             int ntile, float scale) {
    __shared__ float Ks[VT * VLDF], Vs[VT * VLDF];
    const int item = blockIdx.x / ntile, tile = blockIdx.x - item * ntile;
    const int f = item / H, h = item - f * H, C = H * D;
    const int64_t ld = 3 * (int64_t)C;
    const T* base = qkv + (int64_t)f * L * ld + h * D;
    const int tid = threadIdx.x, q = tile * VROWS + tid;
    const int qc = q < L ? q : L - 1;
    const T* orow = o + ((int64_t)f * L + qc) * C + h * D;
    const T* dorow = d_o + ((int64_t)f * L + qc) * C + h * D;
    float qv[D], dov[D], dq[D];
    float Dq = 0.f;
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        float v[4], w[4], u[4];
        ld4(base + (int64_t)qc * ld + c, v); ld4(dorow + c, w); ld4(orow + c, u);
#pragma unroll
        for (int j = 0; j < 4; ++j) { qv[c + j] = v[j]; dov[c + j] = w[j]; Dq = fmaf(w[j], u[j], Dq); dq[c + j] = 0.f; }
    }
    const float Lq = lse[((int64_t)f * H + h) * L + qc];
    for (int k0 = 0; k0 < L; k0 += VT) {
        __syncthreads();
        stage_tile_f32(Ks, base + C, ld, k0, L, tid);
        stage_tile_f32(Vs, base + 2 * C, ld, k0, L, tid);
        __syncthreads();
        const int kn = (L - k0 < VT) ? L - k0 : VT;
        for (int k = 0; k < kn; ++k) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) { s = fmaf(qv[c], Ks[k * VLDF + c], s); dp = fmaf(dov[c], Vs[k * VLDF + c], dp); }
            const float ds = __expf(s * scale - Lq) * (dp - Dq) * scale;
#pragma unroll
            for (int c = 0; c < D; ++c) dq[c] = fmaf(ds, Ks[k * VLDF + c], dq[c]);
        }
    }
    if (q < L) store_row_acc(dqkv + ((int64_t)f * L + q) * ld + h * D, dq, accumulate);
}

// thread per key; SWEEP 0: dV[k] = sum_q p dO[q]; SWEEP 1: dK[k] = sum_q dS Q[q]  (two sweeps over the query tiles to stay in registers)
template <typename T>
__global__ __launch_bounds__(256) void attn_long_bwd_dkv_valu(const T* __restrict__ qkv, const T* __restrict__ o, const T* __restrict__ d_o,
                                                              const float* __restrict__ lse, T* __restrict__ dqkv, int accumulate, int L, int H,
                                                              int ntile, float scale) {
    __shared__ float Qs[VT * VLDF], dOs[VT * VLDF], Ls[VT], Ds[VT];
    const int item = blockIdx.x / ntile, tile = blockIdx.x - item * ntile;
    const int f = item / H, h = item - f * H, C = H * D;
    const int64_t ld = 3 * (int64_t)C;
    const T* base = qkv + (int64_t)f * L * ld + h * D;
    const T* obase = o + (int64_t)f * L * C + h * D;
    const T* dobase = d_o + (int64_t)f * L * C + h * D;
    const int tid = threadIdx.x, k = tile * VROWS + tid;
    const int kc = k < L ? k : L - 1;
    float kv[D], vv[D], acc[D];
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        float a[4], b[4];
        ld4(base + C + (int64_t)kc * ld + c, a); ld4(base + 2 * C + (int64_t)kc * ld + c, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) { kv[c + j] = a[j]; vv[c + j] = b[j]; }
    }
    for (int sweep = 0; sweep < 2; ++sweep) {
#pragma unroll
        for (int c = 0; c < D; ++c) acc[c] = 0.f;
        for (int q0 = 0; q0 < L; q0 += VT) {
            __syncthreads();
            stage_tile_f32(Qs, base, ld, q0, L, tid);
            stage_tile_f32(dOs, dobase, (int64_t)C, q0, L, tid);
            if (tid < VT) {
                int qi = q0 + tid;
                if (qi > L - 1) qi = L - 1;
                float dsum = 0.f;
#pragma unroll
                for (int c = 0; c < D; c += 4) {
                    float a[4], b[4];
                    ld4(dobase + (int64_t)qi * C + c, a); ld4(obase + (int64_t)qi * C + c, b);
#pragma unroll
                    for (int j = 0; j < 4; ++j) dsum = fmaf(a[j], b[j], dsum);
                }
                Ds[tid] = dsum;
                Ls[tid] = lse[((int64_t)f * H + h) * L + qi];
            }
            __syncthreads();
            const int qn = (L - q0 < VT) ? L - q0 : VT;
            for (int qq = 0; qq < qn; ++qq) {
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int c = 0; c < D; ++c) { s = fmaf(Qs[qq * VLDF + c], kv[c], s); dp = fmaf(dOs[qq * VLDF + c], vv[c], dp); }
                const float p = __expf(s * scale - Ls[qq]);
                const float w = sweep == 0 ? p : p * (dp - Ds[qq]) * scale;
                const float* src = sweep == 0 ? dOs + qq * VLDF : Qs + qq * VLDF;
#pragma unroll
                for (int c = 0; c < D; ++c) acc[c] = fmaf(w, src[c], acc[c]);
            }
        }
        if (k < L) store_row_acc(dqkv + ((int64_t)f * L + k) * ld + (sweep == 0 ? 2 * C : C) + h * D, acc, accumulate);
    }
}

}  // namespace

// entry points used by maed_attn_spatial_{fwd,bwd} (attn_spatial.hip); arguments as there.  *_launch: bf16 MFMA; *_valu_launch: exact VALU
int maed_attn_long_fwd_launch(const void* qkv, void* o, float* lse, int F, int L, int H, float scale, hipStream_t s) {
    const int ntile = (L + WG_ROWS - 1) / WG_ROWS;
    MAED_CHECK_ARG((int64_t)F * H * ntile < (1ll << 31), MAED_ERR_SHAPE, "attn_long_fwd: grid too large");
    hipLaunchKernelGGL(attn_long_fwd_mfma, dim3((unsigned)(F * H * ntile)), dim3(256), 0, s, (const bf16*)qkv, (bf16*)o, lse, L, H, ntile,
                       scale * 1.44269504088896340736f);
    MAED_CHECK_LAUNCH("attn_long_fwd");
    return MAED_OK;
}

int maed_attn_long_bwd_launch(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int accumulate, int F, int L, int H,
                              float scale, hipStream_t s) {
    const int ntile = (L + WG_ROWS - 1) / WG_ROWS;
    MAED_CHECK_ARG((int64_t)F * H * ntile < (1ll << 31), MAED_ERR_SHAPE, "attn_long_bwd: grid too large");
    const dim3 grid((unsigned)(F * H * ntile));
    hipLaunchKernelGGL(attn_long_bwd_dq_mfma, grid, dim3(256), 0, s, (const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse, (bf16*)dqkv, accumulate,
                       L, H, ntile, scale);
    hipLaunchKernelGGL(attn_long_bwd_dkv_mfma, grid, dim3(256), 0, s, (const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse, (bf16*)dqkv, accumulate,
                       L, H, ntile, scale);
    MAED_CHECK_LAUNCH("attn_long_bwd");
    return MAED_OK;
}

int maed_attn_long_fwd_valu_launch(const void* qkv, void* o, float* lse, int F, int L, int H, float scale, int dtype, hipStream_t s) {
    const int ntile = (L + VROWS - 1) / VROWS;
    const dim3 grid((unsigned)(F * H * ntile));
    MAED_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((attn_long_fwd_valu<T>), grid, dim3(256), 0, s, (const T*)qkv, (T*)o, lse, L, H, ntile, scale));
    MAED_CHECK_LAUNCH("attn_long_fwd(valu)");
    return MAED_OK;
}

int maed_attn_long_bwd_valu_launch(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int accumulate, int F, int L,
                                   int H, float scale, int dtype, hipStream_t s) {
    const int ntile = (L + VROWS - 1) / VROWS;
    const dim3 grid((unsigned)(F * H * ntile));
    MAED_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL((attn_long_bwd_dq_valu<T>), grid, dim3(256), 0, s, (const T*)qkv, (const T*)o, (const T*)d_o, lse, (T*)dqkv, accumulate, L, H,
                           ntile, scale);
        hipLaunchKernelGGL((attn_long_bwd_dkv_valu<T>), grid, dim3(256), 0, s, (const T*)qkv, (const T*)o, (const T*)d_o, lse, (T*)dqkv, accumulate, L, H,
                           ntile, scale);
    });
    MAED_CHECK_LAUNCH("attn_long_bwd(valu)");
    return MAED_OK;
}
