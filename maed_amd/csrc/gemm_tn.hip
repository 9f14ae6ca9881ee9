// Weight-gradient GEMM  dW[N,K] += Y[M,N]^T * X[M,K]  (both operands row-major, the reduction runs over ROWS).
// nn.Linear backward (vision_transformer.py:98-111,124-128 through autograd): dW = dY^T X, db = colsum(dY).
//
// The MFMA fragments want 8 consecutive reduction elements per lane, i.e. M-contiguous data, while memory is
// N/K-contiguous.  Instead of materialising transposed copies in HBM (round-1 first version: ~2.7 ms/step of
// transposes), each thread loads an 8(m) x 8(n) block (8 coalesced 16-B loads, 16 lanes cover a 256-B row
// segment), transposes it in registers with v_perm_b32 and writes eight 16-B rows of the M-contiguous LDS image
// (144-B padded rows + XOR slot swizzle: conflict-free ds_write_b128 AND ds_read_b128).  The main loop is the same
// 128x128x64 / 4-wave / 2x2 v_mfma_f32_32x32x16_bf16 structure as gemm_nt with a register prefetch distance of two
// M-tiles; split-M workgroups accumulate with coalesced fp32 atomics straight into the gradient arena.  The
// workgroups of K-tile 0 also produce the bias gradient (column sums of Y).
#include "common.cuh"
#include "gemm_epilogue.cuh"      // xcd_remap
#include "gemm_x3.h"
#include "prof.h"
#include <stdlib.h>

// M-splits of a weight-gradient GEMM with `tiles` output tiles of 128 x 128.  The kernel runs two workgroups per CU; the sweep after the LDS layout fix
// (profiles/r03_tn_split_sweep_after_layout.txt) says: large outputs (>= 32 tiles: qkv, fc1, fc2) want the grid to fill the chip exactly twice -- as many
// splits as fit in 512 workgroups, never more (a 513th workgroup starts a second round: qkv 63.6 us at 480 workgroups, 88.2 at 528) -- while small
// outputs (proj, the convolutions) are cheaper with ~256: every split adds a pass of fp32 atomics over the output and a prologue / epilogue per workgroup.
// MAED_OPT_TN_TARGET_WGS > 0 overrides with a plain target (the sweep knob).
int maed_tn_splits(int tiles) {
    const int target = maed_opt(MAED_OPT_TN_TARGET_WGS);
    if (target > 0) return (target + tiles - 1) / tiles;
    if (tiles >= 32) return 512 / tiles > 0 ? 512 / tiles : 1;
    return (256 + tiles - 1) / tiles;
}
static int tn_remap() { return 1; }     // XCD-aware (split, tile) order: -6...8 % and 292 -> 189 MB of HBM traffic per launch (profiles/r02_pmc)

bool maed_gemm_tn_sk_ok(int64_t M, int N, int K, int64_t ldy, int64_t ldx, int64_t ldw, const void* Y, const void* X, const void* dW);      // gemm_tn_sk.hip
int maed_gemm_tn_sk_launch(const void* Y, int64_t ldy, const void* X, int64_t ldx, int64_t M, int N, int K, float* dW, int64_t ldw, float* dbias, int grid_opt, hipStream_t s);
int maed_sk_cus(void);                                                                                                          // gemm_sk.hip
bool maed_gemm_tn_dma_ok(int64_t M, int N, int K, int64_t ldy, int64_t ldx);                                            // gemm_tn2.hip
int maed_gemm_tn_dma_launch(const void* Y, int64_t ldy, const void* X, int64_t ldx, int64_t M, int N, int K, float* dW, int64_t ldw, float* dbias, int which, hipStream_t stream);
bool maed_conv3x3_wgrad_rows64_ok(int F, int H, int W, int Cin, int Cout);                                            // conv3x3_rows.hip
int maed_conv3x3_wgrad_rows64_launch(const void* dy, const void* x, float* dW, void* scratch, int F, int H, int W, hipStream_t stream);

#define TN_BM 64      // reduction rows per LDS tile
#define TN_LD 72      // LDS row stride (elements): 144 B

__device__ __forceinline__ uint32_t perm_lo(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }  // {b.lo16, a.lo16}
__device__ __forceinline__ uint32_t perm_hi(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }  // {b.hi16, a.hi16}

// CONV = true: the weight gradient of a stride-1 3x3 SAME convolution (resnetv2.py:74-93) as the same TN GEMM over GATHERED X rows:
//   dW[co][tap*Cin + ci] += sum_m dY[m][co] * X[m + shift(tap)][ci] * inside(m, tap)         m = (f, y, x), K = 9*Cin
// a thread's 8 K-columns lie in ONE tap (Cin % 8 == 0), so the tap's row shift is a constant folded into its loop-invariant lane
// offsets; whether the shifted pixel is inside the image comes from a per-pixel 9-bit mask (maed_conv3x3_tapmask, computed once per
// feature-map size), and a masked row loads a page of zeros instead.  Needs M % 64 == 0 (no ragged tile).  Opt-in path.
struct TnConv { const uint16_t* tapmask; const bf16* zero_page; int Cin, Wimg;
                const int* rowtab = nullptr; };   // S2 (stride-2 convolution): X row of the top-left tap of every output pixel (maed_conv3x3_s2_tables)

// S2 = true (with CONV): the weight gradient of the STRIDE-2 3x3 convolution -- output pixel m = (f, oy, ox) pairs with the input rows
// rowtab[m] + ky * Wimg + kx, no longer m + const: the 8 row bases of a thread's block come from the table (two 16-byte loads per M-tile), the tap's shift is still
// a loop-invariant constant, the per-pixel 9-bit mask says which taps fall inside the image.
template <bool CONV, bool S2 = false>
__global__ __launch_bounds__(256, 2) void gemm_tn_mfma_bf16_kernel(const bf16* __restrict__ Y, int64_t ldy, const bf16* __restrict__ X,
                                                                   int64_t ldx, int64_t M, int N, int K, float* __restrict__ dW,
                                                                   int64_t ldw, float* __restrict__ dbias, int tiles_k, int mtiles_per_split,
                                                                   TnConv cv, int remap
#ifdef MAED_GEMM_ABLATE
                                                                   , int ablate    // diagnostic build only: 1 no atomics, 2 no global loads, 4 no MFMA / fragment reads, 8 no transposing LDS stores
#endif
                                                                   ) {
#ifndef MAED_GEMM_ABLATE
    constexpr int ablate = 0;
#endif
    __shared__ __attribute__((aligned(16))) unsigned short lds[2][2][128 * TN_LD];  // [buf][Y^T | X^T][row n|k][m]
    __shared__ float lcs[8][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, hi = lane >> 5;
    // XCD-aware order: the tn*tk workgroups of one M-split read the same rows of Y and X (each Y column tile tk times, each X column
    // tile tn times).  The dispatcher deals consecutive workgroups round-robin over the 8 XCDs (each with its own L2), which scattered a
    // split over all of them: L2-miss traffic 2.7x the algorithmic bytes (profiles/r02_pmc, gemm_tn).  Remapped so that a split's
    // workgroups share an XCD, they march down the same rows together and the re-reads hit its L2.
    const int lin = remap ? xcd_remap((int)(blockIdx.z * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.z)) : (int)(blockIdx.z * gridDim.x + blockIdx.x);
    const int bx = lin % (int)gridDim.x, bz = lin / (int)gridDim.x;
    const int tile_n = bx / tiles_k, tile_k = bx % tiles_k;
    const int n0 = tile_n * 128, k0 = tile_k * 128;
    const int nmt = (int)((M + TN_BM - 1) / TN_BM);
    const int mt_beg = bz * mtiles_per_split;
    int mt_end = mt_beg + mtiles_per_split;
    if (mt_end > nmt) mt_end = nmt;
    if (mt_beg >= mt_end) return;
    // the main loop runs over FULL M-tiles with unconditional loads; the (at most one, globally last) ragged tile is a
    // separate bound-checked step of the workgroup that owns it
    const int n_full = (int)(M / TN_BM);
    const bool own_tail = (mt_end > n_full);                  // block-uniform
    const int mt_endf = own_tail ? n_full : mt_end;           // end of this workgroup's full tiles

    // staging role: waves 0,1 transpose the Y tile, waves 2,3 the X tile; each thread owns an 8(m) x 8(col) block.
    // `side` is wave-uniform (readfirstlane): the source pointer and row stride live in SGPRs and a load is
    // global_load_dwordx4 v, v_off, s[base] with a loop-invariant 32-bit lane offset.  The first version recomputed 64-bit
    // row*ld products and branched on a row bound for every load: ~10 VALU/SALU instructions per MFMA.
    const int side = __builtin_amdgcn_readfirstlane(tid >> 7), st = tid & 127;
    // column chunk (8 cols) nc = consecutive lanes (16 lanes cover a 256-byte row segment: coalesced), row group (8 rows) mg.
    // LDS image: column c of the tile lives in LDS row R(c) = (c & ~31) | ((c & 7) << 2) | ((c >> 3) & 3) -- inside every block of 32 columns the 4 chunks
    // are interleaved -- and its 16-byte slots are XOR-ed with 4 * ((c >> 5) & 1).  The 8 lanes a transposing ds_write_b128 is serviced with (8 consecutive
    // chunks, one j) then hit 8 distinct bank groups, and a fragment = 32 consecutive LDS rows with one constant slot term is conflict-free under the
    // hardware's real ds_read_b128 lane grouping ({0-3,12-15,20-27}, ...: every residue mod 16; 144-byte rows).  The round-1 layout (row = column, slot XOR-ed
    // with (row >> 3) & 7) was conflict-free for the writes only: SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE
    // (profiles/r03_tn_bf16_sq_counters_before_lane_remap.txt).  The epilogue undoes R(): lanes still cover 32 consecutive k per atomic instruction.
    const int nc = st & 15, mg = st >> 4;
    const bf16* src = side ? X : Y;
    const int64_t lds_src = side ? ldx : ldy;
    const int c0 = (side ? k0 : n0) + nc * 8;
    const int cmax = side ? K : N;
    const bool col_ok = c0 < cmax;                            // N, K are multiples of 8 (checked by the launcher); loop-invariant
    unsigned short* my_lds_base = &lds[0][side][0];
    const int wr_off = ((nc >> 2) * 32 + (nc & 3)) * TN_LD + ((mg ^ (((nc >> 2) & 1) << 2)) << 3);   // column nc*8 + j -> LDS row (nc>>2)*32 + 4j + (nc&3): step 4 rows per j
    // the column sums of a Y tile are needed once per (N-tile, M-split): the K-tile that takes them rotates with the split
    // index so the extra VALU work is spread over all workgroups instead of making the K-tile-0 ones stragglers
    const bool bias_blk = (dbias != nullptr) && (tile_k == (int)(bz % tiles_k));   // block-uniform
    const bool do_bias = bias_blk && (side == 0);   // wave-uniform
    float cs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[j] = 0.f;
    // lanes whose columns lie beyond N / K load column 0 instead (always valid), never store, and their LDS rows are zeroed once
    const int ldi = (int)lds_src;
    int tap = 0, col_in = col_ok ? c0 : 0;                    // CONV, X side: K column -> (tap, channel); the tap's pixel shift in rows
    int shift = 0;
    if constexpr (CONV) {
        if (side) { tap = col_in / cv.Cin; col_in -= tap * cv.Cin; shift = S2 ? (tap / 3) * cv.Wimg + (tap % 3) : (tap / 3 - 1) * cv.Wimg + (tap % 3 - 1); }
    }
    const int vo0 = (mg * 8 + shift) * ldi + col_in, vo1 = vo0 + ldi, vo2 = vo1 + ldi, vo3 = vo2 + ldi, vo4 = vo3 + ldi, vo5 = vo4 + ldi,
              vo6 = vo5 + ldi, vo7 = vo6 + ldi;               // fits 32 bits: the launcher checks ld < 2^24
    if (!col_ok) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                *reinterpret_cast<uint4*>(my_lds_base + (size_t)b * (2 * 128 * TN_LD) + wr_off + 4 * j * TN_LD) = make_uint4(0u, 0u, 0u, 0u);
    }

    uint4 r0_0, r0_1, r0_2, r0_3, r0_4, r0_5, r0_6, r0_7, r1_0, r1_1, r1_2, r1_3, r1_4, r1_5, r1_6, r1_7;
#define TN_LD1(S, i, tb_) r##S##_##i = *reinterpret_cast<const uint4*>((tb_) + vo##i);
    // CONV, X side: row i of the 8-row group is taken from the image only if bit `tap` of its pixel's mask is set, else from the zero page
#define TN_LD1M(S, i, tb_, mw_) r##S##_##i = *reinterpret_cast<const uint4*>(((((mw_) >> (((i) & 1) * 16 + tap)) & 1u) ? (tb_) + vo##i : cv.zero_page));
    // S2: row i of the block comes from X row rt_ + shift (rt_: the table's entry for that output pixel)
#define TN_LD1S(S, i, mw_, rt_) r##S##_##i = *reinterpret_cast<const uint4*>(((((mw_) >> (((i) & 1) * 16 + tap)) & 1u) ? src + ((int64_t)(rt_) + shift) * ldi + col_in : cv.zero_page));
#define TN_LOAD(S, mt_) if (!(ablate & 2)) { int mtc__ = (mt_); if (mtc__ > mt_endf - 1) mtc__ = mt_endf - 1; \
        const bf16* tb__ = src + (int64_t)mtc__ * TN_BM * lds_src; \
        bool masked__ = false; \
        if constexpr (CONV) masked__ = side != 0; \
        if (masked__ && S2) { \
            const uint4 mk__ = *reinterpret_cast<const uint4*>(cv.tapmask + (int64_t)mtc__ * TN_BM + mg * 8); \
            const uint4 ra__ = *reinterpret_cast<const uint4*>(cv.rowtab + (int64_t)mtc__ * TN_BM + mg * 8), rb__ = *reinterpret_cast<const uint4*>(cv.rowtab + (int64_t)mtc__ * TN_BM + mg * 8 + 4); \
            TN_LD1S(S, 0, mk__.x, (int)ra__.x) TN_LD1S(S, 1, mk__.x, (int)ra__.y) TN_LD1S(S, 2, mk__.y, (int)ra__.z) TN_LD1S(S, 3, mk__.y, (int)ra__.w) \
            TN_LD1S(S, 4, mk__.z, (int)rb__.x) TN_LD1S(S, 5, mk__.z, (int)rb__.y) TN_LD1S(S, 6, mk__.w, (int)rb__.z) TN_LD1S(S, 7, mk__.w, (int)rb__.w) \
        } else if (masked__) { \
            const uint4 mk__ = *reinterpret_cast<const uint4*>(cv.tapmask + (int64_t)mtc__ * TN_BM + mg * 8); \
            TN_LD1M(S, 0, tb__, mk__.x) TN_LD1M(S, 1, tb__, mk__.x) TN_LD1M(S, 2, tb__, mk__.y) TN_LD1M(S, 3, tb__, mk__.y) \
            TN_LD1M(S, 4, tb__, mk__.z) TN_LD1M(S, 5, tb__, mk__.z) TN_LD1M(S, 6, tb__, mk__.w) TN_LD1M(S, 7, tb__, mk__.w) \
        } else { \
            TN_LD1(S, 0, tb__) TN_LD1(S, 1, tb__) TN_LD1(S, 2, tb__) TN_LD1(S, 3, tb__) TN_LD1(S, 4, tb__) TN_LD1(S, 5, tb__) TN_LD1(S, 6, tb__) TN_LD1(S, 7, tb__) } }
    // ragged tile: rows >= M contribute zeros
#define TN_LD1C(S, i, tb_, mrow0) r##S##_##i = ((mrow0) + mg * 8 + i < M) ? *reinterpret_cast<const uint4*>((tb_) + vo##i) : make_uint4(0u, 0u, 0u, 0u);
#define TN_LOAD_TAIL(S) { const int64_t mrow0__ = (int64_t)n_full * TN_BM; const bf16* tb__ = src + mrow0__ * lds_src; \
        TN_LD1C(S, 0, tb__, mrow0__) TN_LD1C(S, 1, tb__, mrow0__) TN_LD1C(S, 2, tb__, mrow0__) TN_LD1C(S, 3, tb__, mrow0__) \
        TN_LD1C(S, 4, tb__, mrow0__) TN_LD1C(S, 5, tb__, mrow0__) TN_LD1C(S, 6, tb__, mrow0__) TN_LD1C(S, 7, tb__, mrow0__) }
    // transposed 16-B row for column pair d (dword index) half b: rows 0..7 of that column
#define TN_ROW(S, COMP, PERM) make_uint4(PERM(r##S##_0.COMP, r##S##_1.COMP), PERM(r##S##_2.COMP, r##S##_3.COMP), \
                                         PERM(r##S##_4.COMP, r##S##_5.COMP), PERM(r##S##_6.COMP, r##S##_7.COMP))
#define TN_CS(S, COMP, j0) { const uint32_t w__[8] = {r##S##_0.COMP, r##S##_1.COMP, r##S##_2.COMP, r##S##_3.COMP, r##S##_4.COMP, r##S##_5.COMP, r##S##_6.COMP, r##S##_7.COMP}; \
        _Pragma("unroll") for (int i__ = 0; i__ < 8; ++i__) { cs[j0] += __uint_as_float(w__[i__] << 16); cs[j0 + 1] += __uint_as_float(w__[i__] & 0xffff0000u); } }
#define TN_STORE(S, buf_) if (col_ok && !(ablate & 8)) { unsigned short* d__ = my_lds_base + (size_t)(buf_) * (2 * 128 * TN_LD) + wr_off; \
        *reinterpret_cast<uint4*>(d__ + 0 * TN_LD) = TN_ROW(S, x, perm_lo); *reinterpret_cast<uint4*>(d__ + 4 * TN_LD) = TN_ROW(S, x, perm_hi); \
        *reinterpret_cast<uint4*>(d__ + 8 * TN_LD) = TN_ROW(S, y, perm_lo); *reinterpret_cast<uint4*>(d__ + 12 * TN_LD) = TN_ROW(S, y, perm_hi); \
        *reinterpret_cast<uint4*>(d__ + 16 * TN_LD) = TN_ROW(S, z, perm_lo); *reinterpret_cast<uint4*>(d__ + 20 * TN_LD) = TN_ROW(S, z, perm_hi); \
        *reinterpret_cast<uint4*>(d__ + 24 * TN_LD) = TN_ROW(S, w, perm_lo); *reinterpret_cast<uint4*>(d__ + 28 * TN_LD) = TN_ROW(S, w, perm_hi); \
        if (do_bias) { TN_CS(S, x, 0) TN_CS(S, y, 2) TN_CS(S, z, 4) TN_CS(S, w, 6) } }

    f32x16_t acc00, acc01, acc10, acc11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }
    // fragment reads: 32 consecutive LDS rows of one 32-column block, 16-B slot (2*kk + hi) ^ 4 * (block & 1): the second sub-tile of a wave is the odd block
    const int arow0 = wr * 64 + l31, arow1 = arow0 + 32, brow0 = wc * 64 + l31, brow1 = brow0 + 32;
#define TN_COMPUTE(buf_) if (!(ablate & 4)) { const unsigned short* A__ = &lds[buf_][0][0]; const unsigned short* B__ = &lds[buf_][1][0]; \
        _Pragma("unroll") for (int kk = 0; kk < TN_BM / 16; ++kk) { const int sl = 2 * kk + hi; \
            const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(A__ + arow0 * TN_LD + (sl << 3)); \
            const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(A__ + arow1 * TN_LD + ((sl ^ 4) << 3)); \
            const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(B__ + brow0 * TN_LD + (sl << 3)); \
            const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(B__ + brow1 * TN_LD + ((sl ^ 4) << 3)); \
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc00, 0, 0, 0); \
            acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc01, 0, 0, 0); \
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc10, 0, 0, 0); \
            acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc11, 0, 0, 0); } }

    __syncthreads();                            // (zeroed LDS rows of out-of-range columns are in place)
    // (a four-set variant -- loads issued three tile times ahead instead of one -- measured the same or slower: the loop is not waiting on
    // vmcnt; profiles/r02_gemm_tn_ablation.txt: MFMA + LDS transposes alone take 3/4 of the kernel's time)
    if (mt_beg < mt_endf) {
        TN_LOAD(0, mt_beg);
        TN_LOAD(1, mt_beg + 1);
        TN_STORE(0, 0);
        __syncthreads();
        int mt = mt_beg;
        for (; mt + 1 < mt_endf; mt += 2) {
            TN_LOAD(0, mt + 2);
            TN_COMPUTE(0);
            TN_STORE(1, 1);
            __syncthreads();
            TN_LOAD(1, mt + 3);
            TN_COMPUTE(1);
            if (mt + 2 < mt_endf) TN_STORE(0, 0);   // guarded: the column sums must not see a clamped duplicate tile
            __syncthreads();
        }
        if (mt < mt_endf) TN_COMPUTE(0);
    }
    if (own_tail) {                             // block-uniform
        __syncthreads();
        TN_LOAD_TAIL(0);
        TN_STORE(0, 0);
        __syncthreads();
        TN_COMPUTE(0);
    }

    // D[LDS row of n][LDS row of k]: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*hi; LDS row r5 of a 32-column block is column ((r5 & 3) << 3) + (r5 >> 2):
    // the atomics of a half-wave still hit 32 consecutive k (one 128-byte line), in interleaved order
#define TN_UNR(r5_) ((((r5_) & 3) << 3) + ((r5_) >> 2))
#define TN_EPI(acc_, i_, j_) { const int kcol = k0 + wc * 64 + (j_) * 32 + TN_UNR(l31); \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) { const int nrow = n0 + wr * 64 + (i_) * 32 + TN_UNR((r & 3) + 8 * (r >> 2) + 4 * hi); \
            if (nrow < N && kcol < K && !(ablate & 1)) atomicAdd(dW + (int64_t)nrow * ldw + kcol, acc_[r]); } }
    TN_EPI(acc00, 0, 0) TN_EPI(acc01, 0, 1) TN_EPI(acc10, 1, 0) TN_EPI(acc11, 1, 1)

    if (bias_blk) {   // block-uniform branch
        if (side == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) lcs[mg][nc * 8 + j] = cs[j];
        }
        __syncthreads();
        if (tid < 128) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += lcs[g][tid];
            if (n0 + tid < N) atomicAdd(dbias + n0 + tid, s);
        }
    }
}

extern "C" int maed_gemm_tn_wgrad(const void* Y, int64_t ldy, const void* X, int64_t ldx, int64_t M, int N, int K, float* dW, int64_t ldw,
                                  float* dbias, int dtype, void* stream) {
    MAED_CHECK_ARG(Y && X && dW, MAED_ERR_ARG, "gemm_tn_wgrad: null pointer");
    const double es__ = (dtype == MAED_BF16) ? 2.0 : 4.0;
    ProfScope prof__(PROF_TN_ALL, stream, 2.0 * (double)M * N * K, es__ * (double)M * ((double)N + K) + 4.0 * (double)N * K);
    const int np_call = maed_x3_take_dtype(dtype);
    if (dtype == MAED_F32) {        // fp32 operands on the split-bf16 MFMA kernel (gemm_x3.hip)
        const int np = np_call ? np_call : maed_x3_planes();
        MAED_CHECK_ARG(np, MAED_ERR_UNSUPPORTED, "gemm_tn_wgrad: f32 needs the split-bf16 matmul mode (maed_set_option(MAED_OPT_F32_MATMUL, 1 or 2)); the exact mode uses "
                                                 "transposed copies + gemm_nt");
        MAED_CHECK_ARG(M > 0 && N > 0 && K > 0 && N % 4 == 0 && K % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0 && ldw >= K, MAED_ERR_SHAPE,
                       "gemm_tn_wgrad(f32): need N, K, ldy, ldx multiples of 4 (N=%d K=%d)", N, K);
        MAED_CHECK_ARG(is_aligned(Y, 16) && is_aligned(X, 16), MAED_ERR_ALIGN, "gemm_tn_wgrad: Y/X must be 16-B aligned");
        MAED_PROPAGATE(maed_gemm_tn_x3_launch(np, Y, ldy, X, ldx, M, N, K, dW, ldw, dbias, nullptr, 0, (hipStream_t)stream));
        MAED_CHECK_LAUNCH("gemm_tn_wgrad(x3)");
        return MAED_OK;
    }
    MAED_CHECK_ARG(dtype == MAED_BF16, MAED_ERR_ARG, "gemm_tn_wgrad: bad dtype %d", dtype);
    MAED_CHECK_ARG(M > 0 && N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0 && ldy % 8 == 0 && ldx % 8 == 0 && ldw >= K, MAED_ERR_SHAPE,
                   "gemm_tn_wgrad: need N, K, ldy, ldx multiples of 8 (N=%d K=%d)", N, K);
    MAED_CHECK_ARG(is_aligned(Y, 16) && is_aligned(X, 16), MAED_ERR_ALIGN, "gemm_tn_wgrad: Y/X must be 16-B aligned");
    MAED_CHECK_ARG(ldy < (1 << 24) && ldx < (1 << 24), MAED_ERR_SHAPE, "gemm_tn_wgrad: row strides must be < 2^24 elements");
    // operands copied unchanged by LDS-DMA, fragments by transposing LDS reads (gemm_tn2.hip).  In isolation it wins 2-9 % on the STE's linears and loses 20-30 % on
    // the backbone's narrow outputs (profiles/r05_tn_dma_micro.txt); IN SITU -- beside the dy -> dx chain on the side stream, where its copies cost no VALU slots --
    // it is the faster choice for every shape: train step 20.28 / 20.33 -> 20.02 / 20.11 ms same-box (profiles/r05_tn_dma_bench_ab.txt).  Option value 3: its
    // 256 x 256 tile (rejected: see gemm_tn2.hip), 0: this file's kernel.
    // Option value 4 (round 6): by shape -- this file's kernel for the backbone's long-and-narrow products (M >= 65536 rows onto at most
    // 65536 outputs: stage 1 / 2's 1x1 convolutions, where it is 7-19 us faster per launch in isolation, profiles/r05_tn_micro.txt), the LDS-DMA kernel elsewhere.
    // persistent K-stream kernel (gemm_tn_sk.hip, round 6): 256 x 256 tiles, the reduction dealt to one workgroup per CU, slabs + a fixed-order reduce launch
    if (maed_opt(MAED_OPT_TN_SK) && maed_gemm_tn_sk_ok(M, N, K, ldy, ldx, ldw, Y, X, dW) && ((N + 255) / 256) * ((K + 255) / 256) * 2 <= maed_sk_cus()) {
        const int rc = maed_gemm_tn_sk_launch(Y, ldy, X, ldx, M, N, K, dW, ldw, dbias, maed_opt(MAED_OPT_SK_GRID), (hipStream_t)stream);
        if (rc == MAED_OK) { MAED_CHECK_LAUNCH("gemm_tn_wgrad(sk)"); return MAED_OK; }
    }
    int tnd = maed_opt(MAED_OPT_TN_DMA);
    if (tnd == 4) tnd = (M >= 65536 && (int64_t)N * K <= 65536) ? 0 : 1;
    if (tnd && maed_gemm_tn_dma_ok(M, N, K, ldy, ldx)) {
        MAED_PROPAGATE(maed_gemm_tn_dma_launch(Y, ldy, X, ldx, M, N, K, dW, ldw, dbias, tnd, (hipStream_t)stream));
        MAED_CHECK_LAUNCH("gemm_tn_wgrad(dma)");
        return MAED_OK;
    }
    const int tn = (N + 127) / 128, tk = (K + 127) / 128;
    const int nmt = (int)((M + TN_BM - 1) / TN_BM);
    int splits = maed_tn_splits(tn * tk);
    if (splits > (nmt + 3) / 4) splits = (nmt + 3) / 4;      // at least 4 M-tiles per workgroup
    if (splits < 1) splits = 1;
    const int per = (nmt + splits - 1) / splits;
    const int z = (nmt + per - 1) / per;
#ifdef MAED_GEMM_ABLATE
    hipLaunchKernelGGL(gemm_tn_mfma_bf16_kernel<false>, dim3(tn * tk, 1, z), dim3(256), 0, (hipStream_t)stream, (const bf16*)Y, ldy, (const bf16*)X, ldx, M,
                       N, K, dW, ldw, dbias, tk, per, TnConv{nullptr, nullptr, 0, 0}, tn_remap(), maed_opt(MAED_OPT_ABLATE));
#else
    hipLaunchKernelGGL(gemm_tn_mfma_bf16_kernel<false>, dim3(tn * tk, 1, z), dim3(256), 0, (hipStream_t)stream, (const bf16*)Y, ldy, (const bf16*)X, ldx, M,
                       N, K, dW, ldw, dbias, tk, per, TnConv{nullptr, nullptr, 0, 0}, tn_remap());
#endif
    MAED_CHECK_LAUNCH("gemm_tn_wgrad");
    return MAED_OK;
}

// bit t of mask[m] = 1 iff the input pixel (y + t/3 - 1, x + t%3 - 1) of output pixel m = (f, y, x) lies inside the H x W image
__global__ __launch_bounds__(256) void conv_tapmask_kernel(uint16_t* __restrict__ mask, int64_t M, int64_t Mpad, int H, int W) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= Mpad) return;
    unsigned bits = 0;
    if (m < M) {
        const int x = (int)(m % W), y = (int)((m / W) % H);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            bits |= ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W ? 1u : 0u) << t;
        }
    }
    mask[m] = (uint16_t)bits;
}

extern "C" int maed_conv3x3_tapmask(void* mask, int F, int H, int W, void* stream) {
    MAED_CHECK_ARG(mask, MAED_ERR_ARG, "conv3x3_tapmask: null pointer");
    MAED_CHECK_ARG(F > 0 && H > 0 && W > 0, MAED_ERR_SHAPE, "conv3x3_tapmask: bad extents");
    const int64_t M = (int64_t)F * H * W, Mpad = (M + TN_BM - 1) / TN_BM * TN_BM;
    hipLaunchKernelGGL(conv_tapmask_kernel, dim3((unsigned)((Mpad + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (uint16_t*)mask, M, Mpad, H, W);
    MAED_CHECK_LAUNCH("conv3x3_tapmask");
    return MAED_OK;
}

// stride-2 3x3 SAME convolution, per output pixel m = (f, oy, ox): rowtab[m] = X row of its top-left tap ((f H + 2 oy - pad_top) W + 2 ox - pad_left; may lie
// outside the frame -- then the mask keeps it from being read), mask bit t = tap (t / 3, t % 3) inside the H x W image
__global__ __launch_bounds__(256) void conv_s2_tables_kernel(uint16_t* __restrict__ mask, int* __restrict__ rowtab, int64_t M, int64_t Mpad, int H, int W, int Ho,
                                                             int Wo, int pt, int pl) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= Mpad) return;
    unsigned bits = 0;
    int row = 0;
    if (m < M) {
        const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho);
        const int64_t f = m / ((int64_t)Wo * Ho);
        const int y0 = 2 * oy - pt, x0 = 2 * ox - pl;
        row = (int)((f * H + y0) * W + x0);
#pragma unroll
        for (int t = 0; t < 9; ++t) bits |= ((unsigned)(y0 + t / 3) < (unsigned)H && (unsigned)(x0 + t % 3) < (unsigned)W ? 1u : 0u) << t;
    }
    mask[m] = (uint16_t)bits; rowtab[m] = row;
}

extern "C" int maed_conv3x3_s2_tables(void* mask, int* rowtab, int F, int H, int W, int pad_top, int pad_left, int Ho, int Wo, void* stream) {
    MAED_CHECK_ARG(mask && rowtab, MAED_ERR_ARG, "conv3x3_s2_tables: null pointer");
    MAED_CHECK_ARG(F > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && (int64_t)F * H * W < (1ll << 31), MAED_ERR_SHAPE, "conv3x3_s2_tables: bad extents");
    const int64_t M = (int64_t)F * Ho * Wo, Mpad = (M + TN_BM - 1) / TN_BM * TN_BM;
    hipLaunchKernelGGL(conv_s2_tables_kernel, dim3((unsigned)((Mpad + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (uint16_t*)mask, rowtab, M, Mpad, H, W, Ho, Wo,
                       pad_top, pad_left);
    MAED_CHECK_LAUNCH("conv3x3_s2_tables");
    return MAED_OK;
}

// dW (Cout, 3, 3, Cin) += weight gradient of the STRIDE-2 3x3 SAME convolution from dy (F,Ho,Wo,Cout) and x (F,H,W,Cin), channels_last bf16: the TN kernel over
// rows gathered through the tables of maed_conv3x3_s2_tables (computed once per feature-map geometry).  F*Ho*Wo % 64 == 0.
extern "C" int maed_conv3x3_s2_wgrad(const void* dy, const void* x, const void* tapmask, const int* rowtab, const void* zero_page, float* dW, int F, int H, int W,
                                     int Cin, int Cout, int Ho, int Wo, int dtype, void* stream) {
    MAED_CHECK_ARG(dy && x && tapmask && rowtab && zero_page && dW, MAED_ERR_ARG, "conv3x3_s2_wgrad: null pointer");
    MAED_CHECK_ARG(dtype == MAED_BF16, MAED_ERR_UNSUPPORTED, "conv3x3_s2_wgrad: bf16 only (dtype=%d)", dtype);
    const int64_t M = (int64_t)F * Ho * Wo;
    const int N = Cout, K = 9 * Cin;
    ProfScope prof__(PROF_TN_CONV, stream, 2.0 * (double)M * Cout * 9 * Cin, 2.0 * ((double)M * Cout + (double)F * H * W * Cin) + 36.0 * (double)Cout * Cin);
    MAED_CHECK_ARG(M > 0 && M % TN_BM == 0, MAED_ERR_SHAPE, "conv3x3_s2_wgrad: F*Ho*Wo = %lld must be a multiple of 64", (long long)M);
    MAED_CHECK_ARG(Cin % 8 == 0 && Cout % 8 == 0 && Cin < (1 << 20) && W + 1 < (1 << 10) && (int64_t)F * H * W < (1ll << 31), MAED_ERR_SHAPE,
                   "conv3x3_s2_wgrad: need Cin, Cout multiples of 8 (Cin=%d Cout=%d)", Cin, Cout);
    MAED_CHECK_ARG(is_aligned(dy, 16) && is_aligned(x, 16) && is_aligned(tapmask, 16) && is_aligned(rowtab, 16) && is_aligned(zero_page, 16), MAED_ERR_ALIGN, "conv3x3_s2_wgrad: 16-B alignment");
    const int tn = (N + 127) / 128, tk = (K + 127) / 128;
    const int nmt = (int)(M / TN_BM);
    int splits = maed_tn_splits(tn * tk);
    if (splits > (nmt + 3) / 4) splits = (nmt + 3) / 4;
    if (splits < 1) splits = 1;
    const int per = (nmt + splits - 1) / splits;
    const int z = (nmt + per - 1) / per;
    TnConv cv{(const uint16_t*)tapmask, (const bf16*)zero_page, Cin, W};
    cv.rowtab = rowtab;
    hipLaunchKernelGGL((gemm_tn_mfma_bf16_kernel<true, true>), dim3(tn * tk, 1, z), dim3(256), 0, (hipStream_t)stream, (const bf16*)dy, (int64_t)Cout,
                       (const bf16*)x, (int64_t)Cin, M, N, K, dW, (int64_t)K, (float*)nullptr, tk, per, cv, tn_remap()
#ifdef MAED_GEMM_ABLATE
                       , 0
#endif
                       );
    MAED_CHECK_LAUNCH("conv3x3_s2_wgrad");
    return MAED_OK;
}

extern "C" int maed_conv3x3_wgrad(const void* dy, const void* x, const void* tapmask, const void* zero_page, float* dW, int F, int H, int W, int Cin,
                                  int Cout, int dtype, void* stream) {
    MAED_CHECK_ARG(dy && x && tapmask && zero_page && dW, MAED_ERR_ARG, "conv3x3_wgrad: null pointer");
    const double es__ = (dtype == MAED_BF16) ? 2.0 : 4.0;
    ProfScope prof__(PROF_TN_CONV, stream, 2.0 * (double)F * H * W * Cout * 9 * Cin, es__ * (double)F * H * W * ((double)Cout + Cin) + 36.0 * (double)Cout * Cin);
    const int np_call = maed_x3_take_dtype(dtype);
    MAED_CHECK_ARG(dtype == MAED_BF16 || dtype == MAED_F32, MAED_ERR_ARG, "conv3x3_wgrad: bad dtype %d", dtype);
    const int64_t M = (int64_t)F * H * W;
    const int N = Cout, K = 9 * Cin;
    if (dtype == MAED_F32) {        // fp32 operands on the split-bf16 MFMA kernel (gemm_x3.hip)
        const int np = np_call ? np_call : maed_x3_planes();
        MAED_CHECK_ARG(np, MAED_ERR_UNSUPPORTED, "conv3x3_wgrad: f32 needs the split-bf16 matmul mode (maed_set_option(MAED_OPT_F32_MATMUL, 1 or 2))");
        MAED_CHECK_ARG(M > 0 && Cin % 4 == 0 && Cout % 4 == 0, MAED_ERR_SHAPE, "conv3x3_wgrad(f32): Cin / Cout must be multiples of 4 (Cin=%d Cout=%d)", Cin, Cout);
        MAED_CHECK_ARG(is_aligned(dy, 16) && is_aligned(x, 16) && is_aligned(tapmask, 16), MAED_ERR_ALIGN, "conv3x3_wgrad: 16-B alignment");
        const X3TnConv cv{(const uint16_t*)tapmask, Cin, W};
        MAED_PROPAGATE(maed_gemm_tn_x3_launch(np, dy, (int64_t)Cout, x, (int64_t)Cin, M, N, K, dW, (int64_t)K, nullptr, &cv, 0, (hipStream_t)stream));
        MAED_CHECK_LAUNCH("conv3x3_wgrad(x3)");
        return MAED_OK;
    }
    if (maed_conv3x3_wgrad_rows64_ok(F, H, W, Cin, Cout)) {      // 64 -> 64 channels (stage 1): one image row per work item, all nine taps from LDS (conv3x3_rows.hip)
        MAED_CHECK_ARG(is_aligned(dy, 16) && is_aligned(x, 16), MAED_ERR_ALIGN, "conv3x3_wgrad: 16-B alignment");
        return maed_conv3x3_wgrad_rows64_launch(dy, x, dW, nullptr, F, H, W, (hipStream_t)stream);
    }
    MAED_CHECK_ARG(M > 0 && M % TN_BM == 0, MAED_ERR_SHAPE, "conv3x3_wgrad: F*H*W = %lld must be a multiple of 64", (long long)M);
    MAED_CHECK_ARG(Cin % 8 == 0 && Cout % 8 == 0 && Cin < (1 << 20) && W + 1 < (1 << 10), MAED_ERR_SHAPE, "conv3x3_wgrad: need Cin, Cout multiples of 8 (Cin=%d Cout=%d)", Cin, Cout);
    MAED_CHECK_ARG(is_aligned(dy, 16) && is_aligned(x, 16) && is_aligned(tapmask, 16) && is_aligned(zero_page, 16), MAED_ERR_ALIGN, "conv3x3_wgrad: 16-B alignment");
    const int tn = (N + 127) / 128, tk = (K + 127) / 128;
    const int nmt = (int)(M / TN_BM);
    int splits = maed_tn_splits(tn * tk);
    if (splits > (nmt + 3) / 4) splits = (nmt + 3) / 4;
    if (splits < 1) splits = 1;
    const int per = (nmt + splits - 1) / splits;
    const int z = (nmt + per - 1) / per;
    hipLaunchKernelGGL(gemm_tn_mfma_bf16_kernel<true>, dim3(tn * tk, 1, z), dim3(256), 0, (hipStream_t)stream, (const bf16*)dy, (int64_t)Cout,
                       (const bf16*)x, (int64_t)Cin, M, N, K, dW, (int64_t)K, (float*)nullptr, tk, per,
                       TnConv{(const uint16_t*)tapmask, (const bf16*)zero_page, Cin, W}, tn_remap()
#ifdef MAED_GEMM_ABLATE
                       , 0
#endif
                       );
    MAED_CHECK_LAUNCH("conv3x3_wgrad");
    return MAED_OK;
}
