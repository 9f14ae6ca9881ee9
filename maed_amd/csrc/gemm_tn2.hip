// Weight-gradient GEMM  dW[N,K] += Y[M,N]^T * X[M,K]  (both operands row-major, the reduction runs over ROWS) -- round-5 operand path.
// nn.Linear / 1x1-convolution backward (vision_transformer.py:98-111,124-128; resnetv2.py:74-93 kernel 1 through autograd): dW = dY^T X, db = colsum(dY).
//
// gemm_tn.hip transposes every operand tile through registers (8 loads, 32 v_perm, 8 ds_write_b128 per thread and tile) between two barriers: per tile step
// ~3500 cycles for 512 cycles of matrix work, a latency chain (profiles/r02_gemm_tn_ablation.txt, r04_tn_two_teams_rejected.txt).  Here nothing is transposed by
// a thread: both tiles are copied UNCHANGED into LDS by LDS-DMA (row-major [m][128 columns], 256-byte rows, 16-byte chunk c of row m at chunk slot
// c ^ ((m & 3) << 2) -- applied on the SOURCE address, an LDS-DMA writes base + lane * 16) and the MFMA fragments -- 8 reduction rows of one column per lane --
// come out of the row-major image with ds_read_b64_tr_b16 (two reads per fragment; inside a 16-lane group lane 4j + t reads row j, columns 4t .. 4t + 3 and
// receives a column of that 4 x 16 block; the four rows of a read sit in four different 64-byte bank quarters: conflict-free).  The pattern of round 4's
// stem / row-item weight gradients (stem.hip, conv3x3_rows.hip), generalised to any N, K.
// Pipeline: M-tiles of 32 rows (8 KB per operand), a ring of FOUR stages per workgroup (64 KB: two workgroups per CU), copies three tiles ahead,
// ONE barrier per tile:   wait(tile t landed) -> barrier -> issue tile t + 3 into the stage tile t - 1 used -> 8 tr-reads + 4 MFMAs per 16 rows.
// Bias gradient: column sums of Y as one more MFMA against a ones operand in the workgroups that own them (no VALU pass over the tile).
#include "common.cuh"
#include "gemm_epilogue.cuh"      // xcd_remap
#include "prof.h"

#define T2_BM 32                          // reduction rows per stage
#define T2_STAGES 4

// NWR x NWC waves, each RN x RK accumulator tiles of 32 x 32: output tile TN = NWR * RN * 32 rows of dW (columns of Y) by TK = NWC * RK * 32 columns (of X).
//   <2, 2, 2, 2>: 128 x 128, four waves, 64 KB of LDS: two workgroups per CU
//   <2, 4, 4, 2>: 256 x 256, eight waves, 128 KB: one workgroup per CU -- half the LDS fill and half the L2 traffic per MFMA (the 128 x 128 tile moves 512 bytes into
//                 LDS per MFMA: at the CU's 64-128 B/clk of LDS write bandwidth that alone is most of an MFMA's 32 cycles); for outputs of at least 512 x 512
template <int NWR, int NWC, int RN, int RK>
__global__ __launch_bounds__(NWR * NWC * 64, (NWR * NWC) / 4) void gemm_tn_dma_bf16_kernel(const bf16* __restrict__ Y, int64_t ldy, const bf16* __restrict__ X, int64_t ldx,
                                                                                             int64_t M, int N, int K, float* __restrict__ dW, int64_t ldw,
                                                                                             float* __restrict__ dbias, int tiles_k, int mtiles_per_split, int remap) {
    constexpr int NW = NWR * NWC, TN = NWR * RN * 32, TK = NWC * RK * 32;
    constexpr int STAGE_ELEMS = T2_BM * (TN + TK);                     // [Y: 32 m x TN][X: 32 m x TK]
    constexpr int NINSTR = STAGE_ELEMS / 512, IPW = NINSTR / NW;       // 1 KB copy instructions per stage / per wave
    constexpr int YINSTR = T2_BM * TN / 512;
    static_assert(NINSTR % NW == 0 && TN % 128 == 0 && TK % 128 == 0, "tile / wave layout");
    MAED_DYN_SHARED(unsigned short, lds);                              // T2_STAGES * STAGE_ELEMS * 2 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / NWC, wc = wave % NWC, l31 = lane & 31, hi = lane >> 5, i16 = lane & 15;
    // XCD-aware order (see gemm_tn.hip): the workgroups of one M-split share an XCD and march down the same rows together
    const int lin = remap ? xcd_remap((int)(blockIdx.z * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.z)) : (int)(blockIdx.z * gridDim.x + blockIdx.x);
    const int bx = lin % (int)gridDim.x, bz = lin / (int)gridDim.x;
    const int tile_n = bx / tiles_k, tile_k = bx % tiles_k;
    const int n0 = tile_n * TN, k0 = tile_k * TK;
    const int nmt = (int)(M / T2_BM);                         // M % 32 == 0 (launcher)
    const int mt_beg = bz * mtiles_per_split;
    int mt_end = mt_beg + mtiles_per_split;
    if (mt_end > nmt) mt_end = nmt;
    if (mt_beg >= mt_end) return;
    const int nt = mt_end - mt_beg;

    // copies: a stage is NINSTR wave-instructions of 1 KB, the Y tile first; wave w issues IPW consecutive ones.  An instruction covers 1024 / (2 TN) rows of its
    // operand; lane -> (row, 16-byte chunk slot); the global chunk that belongs into that slot undoes the swizzle (slot ^ ((row & 3) << 2)).
    // (columns past N / K -- N, K are multiples of 8, not of the tile -- copy the tile's first chunk instead: every lane takes part in every copy (no divergent
    //  copies: the wave's outstanding-copy count stays exact), the duplicate lands in column slots whose products are never stored)
    uint32_t voff[IPW];
    const char* sbase[IPW]; int64_t sld[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int q = wave * IPW + i, side = q >= YINSTR;
        const int cols = side ? TK : TN, cpr = cols / 8;                  // chunks per row: 16 or 32
        const int qq = side ? q - YINSTR : q, rpi = 64 / cpr;             // rows per instruction: 4 or 2
        const int row = qq * rpi + lane / cpr, slot = lane % cpr, chunk = slot ^ ((row & 3) << 2);
        const int64_t ld = side ? ldx : ldy;
        int c0 = (side ? k0 : n0) + chunk * 8;
        if (c0 >= (side ? K : N)) c0 = side ? k0 : n0;
        voff[i] = (uint32_t)(((int64_t)row * ld + c0) * 2);   // fits 32 bits: ld < 2^24 (launcher)
        sbase[i] = (const char*)(side ? X : Y);
        sld[i] = ld * 2 * T2_BM;                              // bytes per M-tile
    }
#define T2_ISSUE(t_) { const int t__ = (t_); unsigned short* st__ = lds + (t__ & (T2_STAGES - 1)) * STAGE_ELEMS; const int64_t mt__ = mt_beg + t__; \
        _Pragma("unroll") for (int i = 0; i < IPW; ++i) { const int q = wave * IPW + i; \
            MAED_LDS_DMA16(sbase[i] + mt__ * sld[i], voff[i], st__ + q * 512); } }

    f32x16_t acc[RN][RK], accb[RN];
#pragma unroll
    for (int a = 0; a < RN; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[a][r] = 0.f;
#pragma unroll
        for (int b = 0; b < RK; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    }
    // the column sums of a Y tile are needed once per (N-tile, M-split): the K-tile that takes them rotates with the split index
    const bool bias_blk = (dbias != nullptr) && (tile_k == (int)(bz % tiles_k)) && wc == 0;   // wave-uniform
    bf16x8_t ones;
    {
        union { bf16x8_t v; uint32_t u[4]; } o;
        o.u[0] = o.u[1] = o.u[2] = o.u[3] = 0x3f803f80u;
        ones = o.v;
    }
    // transposing reads: reduction row 4 hi + (i16 >> 2) (+ 8) of a 16-row step, column (lane & 16) + 4 (i16 & 3) of a 32-column block;
    // the row's chunk slots are XOR-ed with (row & 3) << 2 = (i16 >> 2) << 2
    const int prow = 4 * hi + (i16 >> 2), pcol = (lane & 16) + 4 * (i16 & 3), psw = (i16 >> 2) << 2;
    int a_off[RN], b_off[RK];
#pragma unroll
    for (int a = 0; a < RN; ++a) { const int c = (wr * RN + a) * 32 + pcol; a_off[a] = prow * TN + (((c >> 3) ^ psw) << 3) + (pcol & 7); }
#pragma unroll
    for (int b = 0; b < RK; ++b) { const int c = (wc * RK + b) * 32 + pcol; b_off[b] = T2_BM * TN + prow * TK + (((c >> 3) ^ psw) << 3) + (pcol & 7); }
#define T2_FRAG(dst_, off_, cols_) { auto lo__ = MAED_DS_READ_TR16(st__ + ks * 16 * (cols_) + (off_)); auto hi__ = MAED_DS_READ_TR16(st__ + (ks * 16 + 8) * (cols_) + (off_)); \
        __builtin_memcpy(&dst_.u[0], &lo__, 8); __builtin_memcpy(&dst_.u[1], &hi__, 8); }

    T2_ISSUE(0);
    if (nt > 1) T2_ISSUE(1);
    if (nt > 2) T2_ISSUE(2);
    // (two copies of the loop, with and without the column-sum MFMAs: a branch inside the tile body would end the scheduling region between the two 16-row steps
    //  and keep the second step's fragment reads from moving under the first step's MFMAs)
#define T2_LOOP(BIAS_) for (int t = 0; t < nt; ++t) { \
        /* tile t has landed once at most the younger tiles' copies (IPW = 4 per wave and tile, issued in order) are outstanding */ \
        const int ahead = nt - 1 - t; \
        if (ahead >= 2) { MAED_WAIT_VMCNT(8); } else if (ahead == 1) { MAED_WAIT_VMCNT(4); } else { MAED_WAIT_VMCNT0(); } \
        __syncthreads();                 /* ... for every wave; and every wave is done with tile t - 1: its stage is free */ \
        if (t + 3 < nt) T2_ISSUE(t + 3); \
        const unsigned short* st__ = lds + (t & (T2_STAGES - 1)) * STAGE_ELEMS; \
        _Pragma("unroll") for (int ks = 0; ks < T2_BM / 16; ++ks) { \
            union { bf16x8_t v; uint2 u[2]; } fa[RN], fb[RK]; \
            _Pragma("unroll") for (int a = 0; a < RN; ++a) T2_FRAG(fa[a], a_off[a], TN) \
            _Pragma("unroll") for (int b = 0; b < RK; ++b) T2_FRAG(fb[b], b_off[b], TK) \
            _Pragma("unroll") for (int a = 0; a < RN; ++a) { \
                _Pragma("unroll") for (int b = 0; b < RK; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a].v, fb[b].v, acc[a][b], 0, 0, 0); \
                if (BIAS_) accb[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a].v, ones, accb[a], 0, 0, 0); \
            } \
        } \
    }
    static_assert(IPW == 4, "the counted waits above assume four copies per wave and tile");
    if (bias_blk) { T2_LOOP(true) } else { T2_LOOP(false) }
#undef T2_LOOP
#undef T2_FRAG
#undef T2_ISSUE

    // D[n][k]: column k = lane & 31, row n = (reg & 3) + 8 (reg >> 2) + 4 hi; a half-wave's atomics cover 32 consecutive k (one 128-byte line)
#pragma unroll
    for (int a = 0; a < RN; ++a)
#pragma unroll
        for (int b = 0; b < RK; ++b) {
            const int kcol = k0 + (wc * RK + b) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nrow = n0 + (wr * RN + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (nrow < N && kcol < K) atomicAdd(dW + (int64_t)nrow * ldw + kcol, acc[a][b][r]);
            }
        }
    if (bias_blk && l31 == 0) {          // every column of the ones product holds the row sums: column 0's lanes add them
#pragma unroll
        for (int a = 0; a < RN; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + (wr * RN + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (n < N) atomicAdd(dbias + n, accb[a][r]);
            }
    }
}

bool maed_gemm_tn_dma_ok(int64_t M, int N, int K, int64_t ldy, int64_t ldx) {
    return M >= T2_BM && M % T2_BM == 0 && N % 8 == 0 && K % 8 == 0 && ldy % 8 == 0 && ldx % 8 == 0 && ldy < (1 << 24) && ldx < (1 << 24)
           && (int64_t)T2_BM * (ldy > ldx ? ldy : ldx) * 2 + 1024 < (1ll << 31);
}

int maed_tn_splits(int tiles);           // gemm_tn.hip

// which: 1 / 2 = 128 x 128 tiles, 3 = 256 x 256 (A/B knob MAED_OPT_TN_DMA; measured, profiles/r05_tn_dma_micro.txt: the big tile LOSES everywhere -- qkv 88 vs 62 us --
// its 252 workgroups end with 256 KB of fp32 atomics each, 64 MB per launch against 31 MB, and one workgroup per CU has nobody to overlap them with)
int maed_gemm_tn_dma_launch(const void* Y, int64_t ldy, const void* X, int64_t ldx, int64_t M, int N, int K, float* dW, int64_t ldw, float* dbias, int which,
                            hipStream_t stream) {
    const bool big = which == 3;
    const int T = big ? 256 : 128;
    const int tn = (N + T - 1) / T, tk = (K + T - 1) / T;
    const int nmt = (int)(M / T2_BM);
    // 128 x 128: two workgroups per CU (gemm_tn.hip's sweep: 512 workgroups for large outputs, ~256 for small ones); 256 x 256: one per CU
    // split sweep of THIS kernel (profiles/r05_tn_split_sweep.txt): the round-3 heuristic holds except for outputs of 32 .. 63 tiles, which want ~384 workgroups
    // instead of 512 (qkv, 48 tiles: 8 splits 56.8 us, 10 splits 63.9; stage-3 strided 3x3 shortcut, 32 tiles: 12 splits 47.3, 16 splits 51.9) -- fewer closing atomics
    const int tiles = tn * tk;
    int splits = big ? (256 + tiles - 1) / tiles : (maed_opt(MAED_OPT_TN_TARGET_WGS) == 0 && tiles >= 32 && tiles < 64) ? 384 / tiles : maed_tn_splits(tiles);
    if (big && tn * tk * splits > 256 && splits > 1) --splits;          // never a second residency round
    if (splits > (nmt + 7) / 8) splits = (nmt + 7) / 8;                 // at least 8 M-tiles (256 rows) per workgroup
    if (splits < 1) splits = 1;
    const int per = (nmt + splits - 1) / splits;
    const int z = (nmt + per - 1) / per;
    if (big) {
        constexpr size_t lds = (size_t)T2_STAGES * T2_BM * 512 * 2;
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_tn_dma_bf16_kernel<2, 4, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
        hipLaunchKernelGGL((gemm_tn_dma_bf16_kernel<2, 4, 4, 2>), dim3(tn * tk, 1, z), dim3(512), lds, stream, (const bf16*)Y, ldy, (const bf16*)X, ldx, M, N, K, dW, ldw, dbias,
                           tk, per, 1);
    } else {
        constexpr size_t lds = (size_t)T2_STAGES * T2_BM * 256 * 2;
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_tn_dma_bf16_kernel<2, 2, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
        hipLaunchKernelGGL((gemm_tn_dma_bf16_kernel<2, 2, 2, 2>), dim3(tn * tk, 1, z), dim3(256), lds, stream, (const bf16*)Y, ldy, (const bf16*)X, ldx, M, N, K, dW, ldw, dbias,
                           tk, per, 1);
    }
    return MAED_OK;
}
