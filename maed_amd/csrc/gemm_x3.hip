// fp32-accurate GEMMs on the bf16 matrix cores ("bf16x3" / "bf16x6" split arithmetic): the f32 parity mode at MFMA speed.
//
// Why: the reference's arithmetic is fp32 throughout (vision_transformer.py:98-111,124-128,146-228; resnetv2.py:74-93) and north_star's
// bar is 1e-3 relative on SMPL parameters.  gfx950 has no TF32; its fp32-input MFMA runs at the VALU rate (157 TFLOP/s, 1/16 of bf16),
// and the exact-f32 VALU kernels of gemm.hip made the parity mode 6x slower than the bf16 mode (139 vs 23 ms per cfg3 train step).
// Every fp32 operand x is written as a sum of NP bf16 numbers, x = x0 + x1 (+ x2) with x0 = bf16(x), x1 = bf16(x - x0), ... (each
// subtraction is exact in fp32), and the product of two operands is the sum of the partial products whose orders add up to less than NP:
//     NP = 2 ("bf16x3"): a0 b0 + a0 b1 + a1 b0             3 MFMAs per product, |error| <= ~2^-16 |a b|  (dropped a1 b1 and the x2 tails)
//     NP = 3 ("bf16x6"): + a0 b2 + a1 b1 + a2 b0           6 MFMAs per product, |error| <= ~2^-23 |a b|  (fp32 level)
// all accumulated in the MFMA's fp32 accumulators.  Peak is 2.5 PFLOP/s / 3 = 833 TFLOP/s (x3) or 417 (x6) against 157 for the fp32 MFMA.
//
// Operands stay fp32 in HBM (the parity mode's activations and standardised weights: 4 B per element, read once): a thread loads 16-byte
// pieces into registers, splits them on the VALU (v_cvt_pk_bf16_f32 + exact residual) and writes the NP planes of the LDS image, so the
// split costs no extra memory pass and -- because staging goes through registers anyway -- gathered rows (implicit-GEMM 3x3 convolution)
// and transposing staging (weight gradients, reduction over rows) come for free: the fragments a MFMA lane needs are picked element by
// element while packing.  LDS image per operand and plane: 128 rows x 32 k (bf16), rows padded to 80 B (20 dwords = 4 x odd:
// conflict-free ds_read_b128 fragments and ds_write_b128 rows).  One LDS buffer, the next K tile's global loads in flight under the
// current tile's MFMAs, LDS-shuffled epilogue as in gemm.hip (all fused epilogues, GroupNorm statistics), XCD-aware tile order.
//
//   gemm_nt_x3_kernel   out = epi(A[M,K] B[N,K]^T)      nn.Linear forward / input gradients, 1x1 convolutions; CONV: 3x3 implicit GEMM
//   gemm_tn_x3_kernel   dW[N,K] += Y[M,N]^T X[M,K]      weight gradients (reduction over rows), CONV: 3x3 weight gradient over gathered rows
#include "common.cuh"
#include "gemm_epilogue.cuh"
#include "gemm_x3.h"

#define X3_BK 32
#define X3_LD 40

namespace {

// one K step (16) of a wave's 2x2 (or 1x2) block of 32x32 tiles from the split LDS planes; TR: accumulators hold the transposed tiles
template <int NP, bool TR, bool HALF_ROWS>
__device__ __forceinline__ void x3_kstep(const unsigned short* Ap, const unsigned short* Bp, int plane_elems, f32x16_t& acc00, f32x16_t& acc01,
                                         f32x16_t& acc10, f32x16_t& acc11) {
    bf16x8_t a0[NP], a1[NP], b0[NP], b1[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        a0[p] = *reinterpret_cast<const bf16x8_t*>(Ap + p * plane_elems);
        b0[p] = *reinterpret_cast<const bf16x8_t*>(Bp + p * plane_elems);
        b1[p] = *reinterpret_cast<const bf16x8_t*>(Bp + p * plane_elems + 32 * X3_LD);
        if constexpr (!HALF_ROWS) a1[p] = *reinterpret_cast<const bf16x8_t*>(Ap + p * plane_elems + 32 * X3_LD);
    }
    // smallest partial products first, the leading a0 b0 last
#pragma unroll
    for (int ord = NP - 1; ord >= 0; --ord)
#pragma unroll
        for (int pa = 0; pa <= ord; ++pa) {
            const int pb = ord - pa;
            if constexpr (TR) {
                acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0[pb], a0[pa], acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1[pb], a0[pa], acc01, 0, 0, 0);
                if constexpr (!HALF_ROWS) {
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0[pb], a1[pa], acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1[pb], a1[pa], acc11, 0, 0, 0);
                }
            } else {
                acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[pa], b0[pb], acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[pa], b1[pb], acc01, 0, 0, 0);
                if constexpr (!HALF_ROWS) {
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[pa], b0[pb], acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[pa], b1[pb], acc11, 0, 0, 0);
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// NT:  out = epi(A[M,K] B[N,K]^T), fp32 operands, K % 32 == 0, 16-byte aligned rows.
// CONV: A rows are gathered pixels of a channels_last image (3x3 taps, TF-SAME zero padding, any stride) and B is addressed as
//       element (n, tap, c) = B[b_base + tap * b_tap + n * b_row + c] -- exactly gemm.hip's conv3x3_glds_bf16_kernel, in fp32.
// NARROW: 128 x 64 output tile (N <= 64: stage 1 of the R50), the four waves take 32 rows each.
// ------------------------------------------------------------------------------------------------------------------------------------
template <int EPI, int NP, bool NARROW, bool CONV, bool GN>
__global__ __launch_bounds__(256, 2) void gemm_nt_x3_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                            int64_t M, int64_t N, int64_t K, int tiles_n, int ktiles_per_split, X3ConvDims d,
                                                            EpiArgs e) {
    constexpr int kPlane = 128 * X3_LD;
    constexpr int kTileElems = 2 * NP * kPlane, kStageElems = 4 * 32 * GL_ST * 2;
    constexpr int kMainElems = kTileElems > kStageElems ? kTileElems : kStageElems;
    __shared__ __attribute__((aligned(16))) unsigned short lds_raw[kMainElems + (GN ? GN_TAB_FLOATS * 2 : 0)];
    double* const gn_tab = reinterpret_cast<double*>(lds_raw + kMainElems);
    if (GN && threadIdx.x < GN_TAB_FLOATS / 2) gn_tab[threadIdx.x] = 0.0;                       // (published by the main loop's barriers)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = NARROW ? wave : wave >> 1, wc = NARROW ? 0 : wave & 1, l31 = lane & 31, hi = lane >> 5;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(id / tiles_n) * 128, n0 = (int64_t)(id % tiles_n) * (NARROW ? 64 : 128);
    const int nkt_total = CONV ? 9 * d.Cin / X3_BK : (int)(K / X3_BK);
    const int kt_beg = blockIdx.z * ktiles_per_split;
    int kt_end = kt_beg + ktiles_per_split;
    if (kt_end > nkt_total) kt_end = nkt_total;
    if (kt_beg >= kt_end) return;

    // staging map: a 128 x 32 fp32 tile = 1024 16-byte pieces, 4 per thread: row (tid >> 3) + 32 i, k offset (tid & 7) * 4 -- 8 lanes cover one
    // 128-byte row segment
    const int srow = tid >> 3, skc = (tid & 7) * 4;
    // buffer loads (common.cuh): resource of the operand + loop-invariant 32-bit lane BYTE offsets + the K tile's uniform byte offset -- no 64-bit lane arithmetic
    // in the loop (the first build spent 16 v_lshl_add_u64 per tile step on it), and an out-of-image tap is a lane offset past the extent (reads zeros).
    // The launchers check that an operand spans < 4 GB.
    uint32_t ao[4], bo[4];
    int iy[4], ix[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = srow + 32 * i;
        const int64_t m = (m0 + row < M) ? m0 + row : M - 1;
        const int64_t br = (n0 + row < N) ? n0 + row : N - 1;
        if constexpr (CONV) {
            const int ox = (int)(m % d.Wo), oy = (int)((m / d.Wo) % d.Ho);
            const int64_t f = m / ((int64_t)d.Wo * d.Ho);
            iy[i] = oy * d.stride - d.pad_top; ix[i] = ox * d.stride - d.pad_left;
            // (top-left tap of a border pixel lies before its image: the offsets carry a constant bias of the padding so that they are non-negative; the tile base
            // below subtracts it again -- base + offset is a valid address whenever the tap is inside the image)
            ao[i] = (uint32_t)((((f * d.H + iy[i]) * d.W + ix[i] + (int64_t)d.pad_top * d.W + d.pad_left) * (int64_t)d.Cin + skc) * 4);
            bo[i] = (uint32_t)((br * d.b_row + skc) * 4);
        } else {
            iy[i] = 0; ix[i] = 0;
            ao[i] = (uint32_t)((m * lda + skc) * 4);
            bo[i] = (uint32_t)((br * ldb + skc) * 4);
        }
    }
    // CONV: the image resource starts `bias` elements in front of the tensor (see ao[]); the weight resource is the whole (Cout, 9, Cin) tensor
    const int64_t a_bias = CONV ? ((int64_t)d.pad_top * d.W + d.pad_left) * d.Cin : 0;
    const maed_buf_t Abuf = maed_make_buf(A - a_bias, CONV ? ((int64_t)d.F * d.H * d.W * d.Cin + a_bias) * 4 : ((M - 1) * lda + K) * 4);
    const maed_buf_t Bbuf = maed_make_buf(B, CONV ? 9 * N * (int64_t)d.Cin * 4 : ((N - 1) * ldb + K) * 4);
    float4 ra[4], rb[4];
    int ty = 0, tx = 0, c0 = 0;                                     // CONV: K tile -> (tap, channel chunk), advanced with the loads
    if constexpr (CONV) { const int k = kt_beg * X3_BK; const int tap = k / d.Cin; ty = tap / 3; tx = tap % 3; c0 = k - tap * d.Cin; }
    auto load_tile = [&](int kt) {
        if constexpr (CONV) {
            const uint32_t ak = (uint32_t)((((int64_t)ty * d.W + tx) * d.Cin + c0) * 4);                      // tap + channel-chunk offset (the bias in ao[] stands for -pad)
            const uint32_t bk = (uint32_t)((d.b_base + (int64_t)(ty * 3 + tx) * d.b_tap + c0) * 4);            // (non-negative for both weight layouts)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = (unsigned)(iy[i] + ty) < (unsigned)d.H && (unsigned)(ix[i] + tx) < (unsigned)d.W;
                ra[i] = maed_buf_load_f4(Abuf, ok ? ao[i] : MAED_BUF_OOB, ak);
                if (!NARROW || i < 2) rb[i] = maed_buf_load_f4(Bbuf, bo[i], bk);
            }
            c0 += X3_BK;
            if (c0 == d.Cin) { c0 = 0; if (++tx == 3) { tx = 0; ++ty; } }
        } else {
            const uint32_t kb = (uint32_t)kt * (X3_BK * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = maed_buf_load_f4(Abuf, ao[i], kb);
                if (!NARROW || i < 2) rb[i] = maed_buf_load_f4(Bbuf, bo[i], kb);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = (srow + 32 * i) * X3_LD + skc;
            uint2 pl[NP];
            split4<NP>(ra[i].x, ra[i].y, ra[i].z, ra[i].w, pl);
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<uint2*>(lds_raw + p * kPlane + off) = pl[p];
            if (!NARROW || i < 2) {
                split4<NP>(rb[i].x, rb[i].y, rb[i].z, rb[i].w, pl);
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<uint2*>(lds_raw + (NP + p) * kPlane + off) = pl[p];
            }
        }
    };

    constexpr bool TR = (EPI != MAED_EPI_ATOMIC_F32);
    f32x16_t acc00, acc01, acc10, acc11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }
    const unsigned short* const Afrag = lds_raw + (wr * (NARROW ? 32 : 64) + l31) * X3_LD + hi * 8;
    const unsigned short* const Bfrag = lds_raw + NP * kPlane + (wc * 64 + l31) * X3_LD + hi * 8;

    load_tile(kt_beg);
    for (int kt = kt_beg; kt < kt_end; ++kt) {
        __syncthreads();                                    // every wave is done with the previous tile's fragments
        store_tile();
        __syncthreads();
        if (kt + 1 < kt_end) load_tile(kt + 1);             // in flight under this tile's MFMAs
#pragma unroll
        for (int kk = 0; kk < X3_BK / 16; ++kk)
            x3_kstep<NP, TR, NARROW>(Afrag + kk * 16, Bfrag + kk * 16, kPlane, acc00, acc01, acc10, acc11);
    }

    if constexpr (!TR) {
        // natural orientation: D[row m][col n], col = lane & 31 -> the 32 lanes of a half-wave hit 32 consecutive columns (coalesced atomics)
#define X3_ATOMIC_EPI(acc_, i_, j_) { const int64_t c = n0 + wc * 64 + (j_) * 32 + l31; \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) { const int64_t row = m0 + wr * (NARROW ? 32 : 64) + (i_) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi; \
            if (row < M && c < N) epilogue_store<EPI, float>(e, row, c, acc_[r]); } }
        X3_ATOMIC_EPI(acc00, 0, 0) X3_ATOMIC_EPI(acc01, 0, 1)
        if constexpr (!NARROW) { X3_ATOMIC_EPI(acc10, 1, 0) X3_ATOMIC_EPI(acc11, 1, 1) }
#undef X3_ATOMIC_EPI
    } else {
        // LDS-shuffled epilogue (gemm.hip): a lane owns one output row in the accumulators; each wave parks its 32 x 64 half-tile in LDS and re-reads
        // it with 8 lanes per row -> 32-byte stores / auxiliary reads, full lines per row
        const bool vec_ok = (e.ldo % 8 == 0) && (e.ldaux % 8 == 0);
        float* stg = reinterpret_cast<float*>(lds_raw) + wave * 32 * GL_ST;
        const int rr = lane >> 3, cc = (lane & 7) * 8;
        const GnTile gnt = GN ? gn_tile(gn_tab, m0, n0, N, e.gn_hw) : GnTile{nullptr, 0, 0, 0};
        GnRegs gnr;
        if constexpr (GN) gn_zero(gnr);
#define X3_SHUFFLE_HALF(accA_, accB_, i_)                                                                              \
        __syncthreads();                                                                                               \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                \
            *reinterpret_cast<float4*>(stg + l31 * GL_ST + 8 * g + 4 * hi) = make_float4(accA_[4 * g], accA_[4 * g + 1], accA_[4 * g + 2], accA_[4 * g + 3]);      \
            *reinterpret_cast<float4*>(stg + l31 * GL_ST + 32 + 8 * g + 4 * hi) = make_float4(accB_[4 * g], accB_[4 * g + 1], accB_[4 * g + 2], accB_[4 * g + 3]); \
        }                                                                                                              \
        __syncthreads();                                                                                               \
        _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                                             \
            const int lr = ps * 8 + rr;                                                                                \
            const int64_t row = m0 + wr * (NARROW ? 32 : 64) + (i_) * 32 + lr, col0 = n0 + wc * 64 + cc;               \
            float v8[8];                                                                                               \
            ld8(stg + lr * GL_ST + cc, v8);                                                                            \
            if (row < M && col0 < N) epilogue_store8<EPI, float>(e, row, col0, N, v8, vec_ok);                         \
            if constexpr (GN) { if (row < M && col0 < N) gn_acc8<float>(gnr, gnt, v8, row); }                          \
        }
        X3_SHUFFLE_HALF(acc00, acc01, 0)
        if constexpr (!NARROW) { X3_SHUFFLE_HALF(acc10, acc11, 1) }
#undef X3_SHUFFLE_HALF
        if constexpr (GN) {
            gn_commit(gnr, gnt, lane, n0 + wc * 64 + cc, N);
            __syncthreads();
            gn_flush(gnt, e.gn_sums, m0, M, e.gn_hw, NARROW ? 64 : 128, tid, 256);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// TN:  dW[N,K] += Y[M,N]^T X[M,K]   (fp32 operands, fp32 atomics, split over M; dbias[N] += colsum(Y), exact fp32)
// The MFMA fragments want 8 consecutive REDUCTION elements per lane while memory is N/K-contiguous: a thread loads an 8 (m) x 4 (column)
// block with eight 16-byte loads and, while splitting, packs each column's 8 m-values into one 16-byte row of the M-contiguous LDS image
// [column][m] -- the transpose is free (v_cvt_pk_bf16_f32 takes any two registers).
// CONV: the weight gradient of a stride-1 3x3 SAME convolution, dW[co][tap*Cin + ci] += sum_m dY[m][co] X[m + shift(tap)][ci] inside(m, tap),
// with the per-pixel 9-bit tap mask of maed_conv3x3_tapmask (a thread's 4 K-columns lie in one tap: Cin % 4 == 0).
// ------------------------------------------------------------------------------------------------------------------------------------
template <int NP, bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_tn_x3_kernel(const float* __restrict__ Y, int64_t ldy, const float* __restrict__ X, int64_t ldx,
                                                            int64_t M, int N, int K, float* __restrict__ dW, int64_t ldw,
                                                            float* __restrict__ dbias, int tiles_k, int mtiles_per_split, X3TnConv cv) {
    constexpr int kPlane = 128 * X3_LD;
    __shared__ __attribute__((aligned(16))) unsigned short lds[2 * NP * kPlane];     // [Y^T | X^T][plane][column][m]
    __shared__ float lcs[4][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, hi = lane >> 5;
    // XCD-aware order: the workgroups of one M-split read the same rows of Y and X -> one XCD, its L2 serves the re-reads (gemm_tn.hip)
    const int lin = xcd_remap((int)(blockIdx.z * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.z));
    const int bx = lin % (int)gridDim.x, bz = lin / (int)gridDim.x;
    const int tile_n = bx / tiles_k, tile_k = bx % tiles_k;
    const int n0 = tile_n * 128, k0 = tile_k * 128;
    const int nmt = (int)((M + X3_BK - 1) / X3_BK);
    const int mt_beg = bz * mtiles_per_split;
    int mt_end = mt_beg + mtiles_per_split;
    if (mt_end > nmt) mt_end = nmt;
    if (mt_beg >= mt_end) return;

    // staging role: threads 0..127 transpose the Y tile, 128..255 the X tile; a thread owns rows mg*8 .. +7 and 4 columns.  CONSECUTIVE lanes take consecutive
    // 16-byte column chunks of a row (32 lanes = 512 contiguous bytes): the first version gave consecutive lanes different row groups, i.e. one memory request
    // per 16 bytes, and ran 3.2x the bf16 kernel's time.  LDS image: column c of the tile lives in LDS row R(c) = (c & ~31) | ((c & 3) << 3) | ((c >> 2) & 7)
    // (the 8 chunks of a 32-column block interleaved): the 8 lanes a ds_write_b128 is serviced with write 8 consecutive rows (5 slots per row: 8 distinct
    // bank groups), a fragment is 32 consecutive rows (conflict-free), and the epilogue undoes R() -- a half-wave's atomics still cover 32 consecutive k.
    const int side = __builtin_amdgcn_readfirstlane(tid >> 7), st = tid & 127;
    const int nc = st & 31, mg = st >> 5;
    const float* src = side ? X : Y;
    const int64_t ld = side ? ldx : ldy;
    const int c0 = (side ? k0 : n0) + nc * 4;
    const bool col_ok = c0 < (side ? K : N);                  // N, K are multiples of 4 (launcher)
    int tap = 0, col_in = col_ok ? c0 : 0, shift = 0;
    if constexpr (CONV) {
        if (side) { tap = col_in / cv.Cin; col_in -= tap * cv.Cin; shift = (tap / 3 - 1) * cv.Wimg + (tap % 3 - 1); }
    }
    // buffer loads (common.cuh): one resource per side; the lane's BYTE offset of row j of tile mt = row_off[j] + mt * tile_bytes is range-checked by the hardware
    // against the tensor's extent, so rows past M read zeros by themselves (a ragged last tile needs no predicate) and a masked tap or a column past N / K is the
    // out-of-range offset.  CONV: a tap shifts the X rows by up to Wimg + 1 rows either way -- the resource starts that many rows in front of the tensor.
    const int64_t row_bias = (CONV && side) ? (int64_t)cv.Wimg + 1 : 0;
    const maed_buf_t buf = maed_make_buf(src - row_bias * ld, ((M - 1 + row_bias) * ld + (side ? (CONV ? cv.Cin : K) : N)) * 4);
    const uint32_t lane_off0 = (uint32_t)((((int64_t)(mg * 8 + shift) + row_bias) * ld + col_in) * 4);
    const uint32_t row_bytes = (uint32_t)(ld * 4), tile_bytes = (uint32_t)(X3_BK * ld * 4);
    unsigned short* const my_lds = lds + side * NP * kPlane + ((nc >> 3) * 32 + (nc & 7)) * X3_LD + mg * 8;     // column nc*4 + jj -> LDS row (nc>>3)*32 + 8 jj + (nc&7)
    const bool bias_blk = (dbias != nullptr) && (tile_k == (int)(bz % tiles_k));   // block-uniform: the K tile that takes the column sums rotates
    const bool do_bias = bias_blk && (side == 0);
    float cs[4] = {0.f, 0.f, 0.f, 0.f};

    float4 r[8];
    auto load_tile = [&](int mt) {
        const uint32_t t_off = lane_off0 + (uint32_t)mt * tile_bytes;     // (in the lane offset, not the uniform one: the range check sees it)
        uint32_t mw[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        bool masked = false;
        if constexpr (CONV) masked = side != 0;
        if (masked) { const uint4 mk = *reinterpret_cast<const uint4*>(cv.tapmask + (int64_t)mt * X3_BK + mg * 8); mw[0] = mk.x; mw[1] = mk.y; mw[2] = mk.z; mw[3] = mk.w; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint32_t off = col_ok ? t_off + (uint32_t)j * row_bytes : MAED_BUF_OOB;
            if (masked) off = ((mw[j >> 1] >> ((j & 1) * 16 + tap)) & 1u) ? off : MAED_BUF_OOB;
            r[j] = maed_buf_load_f4(buf, off, 0u);
        }
    };
    auto store_tile = [&]() {
#define X3_TN_COL(COMP, jj_) { \
            uint2 p0[NP], p1[NP]; \
            split4<NP>(r[0].COMP, r[1].COMP, r[2].COMP, r[3].COMP, p0); split4<NP>(r[4].COMP, r[5].COMP, r[6].COMP, r[7].COMP, p1); \
            _Pragma("unroll") for (int p = 0; p < NP; ++p) \
                *reinterpret_cast<uint4*>(my_lds + p * kPlane + 8 * (jj_) * X3_LD) = make_uint4(p0[p].x, p0[p].y, p1[p].x, p1[p].y); \
            if (do_bias) cs[jj_] += ((r[0].COMP + r[1].COMP) + (r[2].COMP + r[3].COMP)) + ((r[4].COMP + r[5].COMP) + (r[6].COMP + r[7].COMP)); }
        X3_TN_COL(x, 0) X3_TN_COL(y, 1) X3_TN_COL(z, 2) X3_TN_COL(w, 3)
#undef X3_TN_COL
    };

    f32x16_t acc00, acc01, acc10, acc11;
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc00[q] = 0.f; acc01[q] = 0.f; acc10[q] = 0.f; acc11[q] = 0.f; }
    const unsigned short* const Afrag = lds + (wr * 64 + l31) * X3_LD + hi * 8;
    const unsigned short* const Bfrag = lds + NP * kPlane + (wc * 64 + l31) * X3_LD + hi * 8;

    load_tile(mt_beg);
    for (int mt = mt_beg; mt < mt_end; ++mt) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (mt + 1 < mt_end) load_tile(mt + 1);
#pragma unroll
        for (int kk = 0; kk < X3_BK / 16; ++kk)
            x3_kstep<NP, false, false>(Afrag + kk * 16, Bfrag + kk * 16, kPlane, acc00, acc01, acc10, acc11);
    }

    // D[LDS row of n][LDS row of k]: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*hi; LDS row r5 of a 32-column block is column ((r5 & 7) << 2) + (r5 >> 3):
    // the atomics of a half-wave hit 32 consecutive k (one 128-byte line), in interleaved order
#define X3_UNR(r5_) ((((r5_) & 7) << 2) + ((r5_) >> 3))
#define X3_TN_EPI(acc_, i_, j_) { const int kcol = k0 + wc * 64 + (j_) * 32 + X3_UNR(l31); \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) { const int nrow = n0 + wr * 64 + (i_) * 32 + X3_UNR((q & 3) + 8 * (q >> 2) + 4 * hi); \
            if (nrow < N && kcol < K) atomicAdd(dW + (int64_t)nrow * ldw + kcol, acc_[q]); } }
    X3_TN_EPI(acc00, 0, 0) X3_TN_EPI(acc01, 0, 1) X3_TN_EPI(acc10, 1, 0) X3_TN_EPI(acc11, 1, 1)
#undef X3_TN_EPI

    if (bias_blk) {   // block-uniform branch
        if (side == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) lcs[mg][nc * 4 + j] = cs[j];
        }
        __syncthreads();
        if (tid < 128) {
            const float s = (lcs[0][tid] + lcs[1][tid]) + (lcs[2][tid] + lcs[3][tid]);
            if (n0 + tid < N) atomicAdd(dbias + n0 + tid, s);
        }
    }
}

}  // namespace

// ---- launchers (C++ linkage; called by the extern "C" entry points of gemm.hip / gemm_tn.hip) -----------------------------------------------

bool maed_x3_nt_shape_ok(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t K) {
    return K % X3_BK == 0 && lda % 4 == 0 && ldb % 4 == 0 && is_aligned(A, 16) && is_aligned(B, 16);
}
// buffer loads with 32-bit byte offsets: every operand must span less than 4 GB
static bool x3_fits32(int64_t rows_a, int64_t lda, int64_t rows_b, int64_t ldb) {
    return (uint64_t)rows_a * (uint64_t)lda * 4 < 0xfffffff0ull && (uint64_t)rows_b * (uint64_t)ldb * 4 < 0xfffffff0ull;
}

template <int EPI, int NP, bool NARROW, bool CONV, bool GN>
static void launch_nt(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const X3ConvDims& d, const EpiArgs& e,
                      int splitk, hipStream_t s) {
    const int tm = (int)((M + 127) / 128), tn = NARROW ? 1 : (int)((N + 127) / 128);
    const int nkt = CONV ? 9 * d.Cin / X3_BK : (int)(K / X3_BK);
    int kps = (nkt + splitk - 1) / splitk;
    if (kps < 1) kps = 1;
    const int z = (nkt + kps - 1) / kps;
    hipLaunchKernelGGL((gemm_nt_x3_kernel<EPI, NP, NARROW, CONV, GN>), dim3((unsigned)(tm * tn), 1, (unsigned)z), dim3(256), 0, s, A, lda, B, ldb, M, N, K, tn, kps,
                       d, e);
}

template <int EPI>
static void launch_nt_np(int np, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const EpiArgs& e, int splitk,
                         hipStream_t s) {
    const X3ConvDims d{};
    if (np == 3) launch_nt<EPI, 3, false, false, false>(A, lda, B, ldb, M, N, K, d, e, splitk, s);
    else if (np == 1) launch_nt<EPI, 1, false, false, false>(A, lda, B, ldb, M, N, K, d, e, splitk, s);     // "bf16x1": one plane, one MFMA (backward products of the mixed mode)
    else launch_nt<EPI, 2, false, false, false>(A, lda, B, ldb, M, N, K, d, e, splitk, s);
}

int maed_gemm_nt_x3_launch(int epilogue, int np, const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const EpiArgs& e,
                           int splitk, hipStream_t s) {
    const float* a = (const float*)A; const float* b = (const float*)B;
    MAED_CHECK_ARG(x3_fits32(M, lda, N, ldb), MAED_ERR_SHAPE, "gemm_nt(x3): an operand larger than 4 GB");
    switch (epilogue) {
        case MAED_EPI_STORE: launch_nt_np<MAED_EPI_STORE>(np, a, lda, b, ldb, M, N, K, e, 1, s); break;
        case MAED_EPI_GELU: launch_nt_np<MAED_EPI_GELU>(np, a, lda, b, ldb, M, N, K, e, 1, s); break;
        case MAED_EPI_RESID_F32: launch_nt_np<MAED_EPI_RESID_F32>(np, a, lda, b, ldb, M, N, K, e, 1, s); break;
        case MAED_EPI_MUL_DGELU: launch_nt_np<MAED_EPI_MUL_DGELU>(np, a, lda, b, ldb, M, N, K, e, 1, s); break;
        case MAED_EPI_ATOMIC_F32: launch_nt_np<MAED_EPI_ATOMIC_F32>(np, a, lda, b, ldb, M, N, K, e, splitk, s); break;
        case MAED_EPI_STORE_F32: launch_nt_np<MAED_EPI_STORE_F32>(np, a, lda, b, ldb, M, N, K, e, 1, s); break;
        case MAED_EPI_TANH: launch_nt_np<MAED_EPI_TANH>(np, a, lda, b, ldb, M, N, K, e, 1, s); break;
        case MAED_EPI_ADD: launch_nt_np<MAED_EPI_ADD>(np, a, lda, b, ldb, M, N, K, e, 1, s); break;
        default: maed_set_error("gemm_nt(x3): bad epilogue %d", epilogue); return MAED_ERR_ARG;
    }
    return MAED_OK;
}

// 1x1 convolution: STORE epilogue (+ GroupNorm statistics of the stored fp32 output)
int maed_conv1x1_x3_launch(int np, const void* x, int64_t ldx, const void* w, int64_t ldw, int64_t M, int Cout, int Cin, const EpiArgs& e, bool gn, hipStream_t s) {
    const X3ConvDims d{};
    const float* a = (const float*)x; const float* b = (const float*)w;
    MAED_CHECK_ARG(x3_fits32(M, ldx, Cout, ldw), MAED_ERR_SHAPE, "conv1x1(x3): an operand larger than 4 GB");
    const bool narrow = Cout <= 64;
#define X3_C1(NP_, NARROW_, GN_) launch_nt<MAED_EPI_STORE, NP_, NARROW_, false, GN_>(a, ldx, b, ldw, M, Cout, Cin, d, e, 1, s)
    if (np == 3) { if (gn) { if (narrow) X3_C1(3, true, true); else X3_C1(3, false, true); } else { if (narrow) X3_C1(3, true, false); else X3_C1(3, false, false); } }
    else { if (gn) { if (narrow) X3_C1(2, true, true); else X3_C1(2, false, true); } else { if (narrow) X3_C1(2, true, false); else X3_C1(2, false, false); } }
#undef X3_C1
    return MAED_OK;
}

// 3x3 implicit GEMM: STORE (+ GN) or ADD epilogue
int maed_conv3x3_x3_launch(int np, const void* x, const void* w, const X3ConvDims& d, int64_t M, int Cout, const EpiArgs& e, bool add, bool gn, hipStream_t s) {
    const float* a = (const float*)x; const float* b = (const float*)w;
    MAED_CHECK_ARG(x3_fits32((int64_t)d.F * d.H * d.W + (int64_t)d.pad_top * d.W + d.pad_left, d.Cin, 9 * (int64_t)Cout, d.Cin), MAED_ERR_SHAPE, "conv3x3(x3): an operand larger than 4 GB");
    const bool narrow = Cout <= 64;
#define X3_C3(EPI_, NP_, NARROW_, GN_) launch_nt<EPI_, NP_, NARROW_, true, GN_>(a, 0, b, 0, M, Cout, 9 * (int64_t)d.Cin, d, e, 1, s)
#define X3_C3_NP(NP_) \
    if (add) { if (narrow) X3_C3(MAED_EPI_ADD, NP_, true, false); else X3_C3(MAED_EPI_ADD, NP_, false, false); } \
    else if (gn) { if (narrow) X3_C3(MAED_EPI_STORE, NP_, true, true); else X3_C3(MAED_EPI_STORE, NP_, false, true); } \
    else { if (narrow) X3_C3(MAED_EPI_STORE, NP_, true, false); else X3_C3(MAED_EPI_STORE, NP_, false, false); }
    if (np == 3) { X3_C3_NP(3) } else if (np == 1 && !gn) { X3_C3_NP(1) } else { X3_C3_NP(2) }      // (np = 1 with statistics: no such caller -- two planes, never less accurate than asked)
#undef X3_C3_NP
#undef X3_C3
    return MAED_OK;
}

int maed_gemm_tn_x3_launch(int np, const void* Y, int64_t ldy, const void* X, int64_t ldx, int64_t M, int N, int K, float* dW, int64_t ldw, float* dbias,
                           const X3TnConv* conv, int target_wgs, hipStream_t s) {
    // 32-bit byte offsets against a buffer extent clamped to 4 GB (as the NT / convolution launchers): a larger operand would wrap into range and read wrong rows.
    // The gathered X rows of the 3x3 weight gradient reach one image row + one pixel past either end of the tensor (masked taps): counted in.
    const int64_t x_rows = M + (conv ? 2 * ((int64_t)conv->Wimg + 1) : 0);
    MAED_CHECK_ARG(x3_fits32(M, ldy, x_rows, ldx), MAED_ERR_SHAPE, "gemm_tn(x3): an operand larger than 4 GB (M=%lld ldy=%lld ldx=%lld): split the batch",
                   (long long)M, (long long)ldy, (long long)ldx);
    const int tn = (N + 127) / 128, tk = (K + 127) / 128;
    const int nmt = (int)((M + X3_BK - 1) / X3_BK);
    // large outputs: fill the chip exactly twice like the bf16 kernel (maed_tn_splits); small ones keep ~384 workgroups -- this kernel runs three
    // workgroups per CU and measured slower with the bf16 kernel's 256 (profiles/r03_x3_tn_split_ab.txt: proj 73 vs 93 us, 56x56 1x1 148 vs 189 us)
    const int tiles = tn * tk;
    int splits = target_wgs > 0 ? (target_wgs + tiles - 1) / tiles : tiles >= 32 ? maed_tn_splits(tiles) : (384 + tiles - 1) / tiles;
    if (splits > (nmt + 7) / 8) splits = (nmt + 7) / 8;      // at least 8 M-tiles (256 rows) per workgroup
    if (splits < 1) splits = 1;
    const int per = (nmt + splits - 1) / splits;
    const int z = (nmt + per - 1) / per;
    const dim3 grid(tn * tk, 1, z);
    const X3TnConv cv = conv ? *conv : X3TnConv{nullptr, 0, 0};
#define X3_TN(NP_, CONV_) hipLaunchKernelGGL((gemm_tn_x3_kernel<NP_, CONV_>), grid, dim3(256), 0, s, (const float*)Y, ldy, (const float*)X, ldx, M, N, K, dW, ldw, \
                                             dbias, tk, per, cv)
    if (np == 3) { if (conv) X3_TN(3, true); else X3_TN(3, false); }
    else if (np == 1) { if (conv) X3_TN(1, true); else X3_TN(1, false); }
    else { if (conv) X3_TN(2, true); else X3_TN(2, false); }
#undef X3_TN
    return MAED_OK;
}
