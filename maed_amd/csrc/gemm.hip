// nn.Linear family: out = epilogue(A[M,K] * B[N,K]^T).  Both operands are K-contiguous (activations
// are row-major, nn.Linear weights are stored (out,in)), which is exactly the MFMA A/B fragment order.
//
//  * bf16: 128x128x64 tile, 4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16 tiles,
//    fp32 accumulation.  Global -> VGPR -> LDS staging (16-byte accesses, full 128-B lines per row),
//    LDS rows padded to 144 B so ds_read_b128 fragment reads are bank-conflict free, double-buffered
//    LDS with the next tile's global loads in flight under the current tile's MFMAs (one barrier per
//    K tile).  Workgroup ids are remapped so the N-tiles of one M-panel run on one XCD (A panel is
//    fetched once per XCD L2; the weight matrix is small and L2/MALL resident).
//  * f32 (parity mode): 64x64x16 LDS-tiled VALU kernel, exact fp32 FMA chains.
// Split-K (grid.z) is available for the atomic weight-gradient epilogue.
#include "common.cuh"
#include "gemm_epilogue.cuh"
#include "gemm_x3.h"

// ------------------------------------------------------------------------------------------------
// VALU kernel (any T; used for f32 and as the cross-check for the MFMA kernel)
// ------------------------------------------------------------------------------------------------
template <int EPI, typename T>
__global__ __launch_bounds__(256) void gemm_nt_valu_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ B,
                                                           int64_t ldb, int64_t M, int64_t N, int64_t K, int64_t k_per_split,
                                                           EpiArgs e) {
    __shared__ float As[16][64 + 1];
    __shared__ float Bs[16][64 + 1];
    const int tid = threadIdx.x;
    const int64_t m0 = (int64_t)blockIdx.y * 64, n0 = (int64_t)blockIdx.x * 64;
    const int64_t kbeg = (int64_t)blockIdx.z * k_per_split;
    const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
    const int ty = tid >> 4, tx = tid & 15;
    const int lr = tid >> 2, lk = (tid & 3) * 4;  // staging: row lr, k offset lk..lk+3
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const int64_t ar = (m0 + lr < M) ? m0 + lr : M - 1;
    const int64_t br = (n0 + lr < N) ? n0 + lr : N - 1;
    // a thread stages 4 consecutive k of one row per operand: one 16-byte load when the rows allow it, and the NEXT K step's loads are issued
    // before this step's FMAs (a step used to be one exposed global-load latency: 3 us x 16 steps for the decoder head's GEMMs)
    const bool vec = sizeof(T) == 4 && (lda % 4 == 0) && (ldb % 4 == 0) && (((uintptr_t)A | (uintptr_t)B) & 15) == 0 && (kbeg % 4 == 0);   // block-uniform
    float va[4], vb[4];
    auto fetch = [&](int64_t k0) {
        const int64_t k = k0 + lk;
        if (vec && k + 4 <= kend) {
            const float4 a4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(A) + ar * lda + k);
            const float4 b4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(B) + br * ldb + k);
            va[0] = a4.x; va[1] = a4.y; va[2] = a4.z; va[3] = a4.w; vb[0] = b4.x; vb[1] = b4.y; vb[2] = b4.z; vb[3] = b4.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                va[i] = (k + i < kend) ? ldf(A + ar * lda + k + i) : 0.f;
                vb[i] = (k + i < kend) ? ldf(B + br * ldb + k + i) : 0.f;
            }
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int64_t k0 = kbeg; k0 < kend; k0 += 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { As[lk + i][lr] = va[i]; Bs[lk + i][lr] = vb[i]; }
        __syncthreads();
        if (k0 + 16 < kend) fetch(k0 + 16);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = m0 + ty * 4 + i, c = n0 + tx * 4 + j;
            if (r < M && c < N) epilogue_store<EPI, T>(e, r, c, acc[i][j]);
        }
}

// ------------------------------------------------------------------------------------------------
// MFMA bf16 kernel
// ------------------------------------------------------------------------------------------------
#define GM_BM 128
#define GM_BN 128
#define GM_BK 64
#define GM_LD 72  // padded LDS row (elements): 144 B = 9 x 16-B slots, 9 coprime with 16 -> conflict-free b128

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_mfma_bf16_kernel(const bf16* __restrict__ A, int64_t lda,
                                                                const bf16* __restrict__ B, int64_t ldb, int64_t M,
                                                                int64_t N, int64_t K, int tiles_n, int ktiles_per_split,
                                                                EpiArgs e) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[2][2][GM_BM * GM_LD];  // [buf][A|B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    const int nwg = gridDim.x;
    const int id = xcd_remap(blockIdx.x, nwg);
    const int64_t m0 = (int64_t)(id / tiles_n) * GM_BM, n0 = (int64_t)(id % tiles_n) * GM_BN;
    const int nkt_total = (int)(K / GM_BK);
    const int kt_beg = blockIdx.z * ktiles_per_split;
    int kt_end = kt_beg + ktiles_per_split;
    if (kt_end > nkt_total) kt_end = nkt_total;
    if (kt_beg >= kt_end) return;

    // staging map: 128 rows x 64 cols = 1024 16-byte chunks per operand, 4 per thread
    // chunk c = tid + 256*i : row = c >> 3, col chunk = c & 7  (8 lanes cover one 128-B row segment)
#define GM_PTRS(i)                                                                            \
    const bf16* ap##i; const bf16* bp##i; int soff##i;                                        \
    {                                                                                         \
        const int c = tid + 256 * i, row = c >> 3, cc = (c & 7) * 8;                          \
        const int64_t ar = (m0 + row < M) ? m0 + row : M - 1;                                 \
        const int64_t br = (n0 + row < N) ? n0 + row : N - 1;                                 \
        ap##i = A + ar * lda + cc; bp##i = B + br * ldb + cc; soff##i = row * GM_LD + cc;     \
    }
    GM_PTRS(0) GM_PTRS(1) GM_PTRS(2) GM_PTRS(3)
    // two register sets = global prefetch distance of TWO K tiles (K is short here: 8..32 tiles per output tile, so
    // HBM/L2 latency, not MFMA issue, decides; with 2 workgroups per CU this keeps 4 tiles in flight per CU)
    // (named scalars, not arrays: the register sets must never be demoted to scratch)
    uint4 ra0_0, ra0_1, ra0_2, ra0_3, rb0_0, rb0_1, rb0_2, rb0_3, ra1_0, ra1_1, ra1_2, ra1_3, rb1_0, rb1_1, rb1_2, rb1_3;
#define GM_LD1(S, i, k0__) ra##S##_##i = *reinterpret_cast<const uint4*>(ap##i + k0__); rb##S##_##i = *reinterpret_cast<const uint4*>(bp##i + k0__);
#define GM_LOAD_TILE(S, kt_)                                                      \
    {                                                                             \
        int ktc__ = (kt_); if (ktc__ > kt_end - 1) ktc__ = kt_end - 1;            \
        const int64_t k0__ = (int64_t)ktc__ * GM_BK;                              \
        GM_LD1(S, 0, k0__) GM_LD1(S, 1, k0__) GM_LD1(S, 2, k0__) GM_LD1(S, 3, k0__) \
    }
#define GM_ST1(S, i, buf_) *reinterpret_cast<uint4*>(&lds[buf_][0][soff##i]) = ra##S##_##i; *reinterpret_cast<uint4*>(&lds[buf_][1][soff##i]) = rb##S##_##i;
#define GM_STORE_TILE(S, buf_) { GM_ST1(S, 0, buf_) GM_ST1(S, 1, buf_) GM_ST1(S, 2, buf_) GM_ST1(S, 3, buf_) }

    // TR: accumulators hold the TRANSPOSED 32x32 tiles (A operand = weight rows n, B operand = activation rows m): a lane
    // then owns one output row m and 4 consecutive columns per register group -> 8/16-byte epilogue accesses.
    // The atomic (weight-gradient) epilogue keeps the natural orientation: for one register the 32 lanes of a half-wave
    // hit 32 CONSECUTIVE columns of one row, which the L2 atomic unit coalesces (row-strided atomics ran 4.5x slower).
    constexpr bool TR = (EPI != MAED_EPI_ATOMIC_F32);
    f32x16_t acc00, acc01, acc10, acc11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }

#define GM_COMPUTE_TILE(buf_)                                                                              \
    {                                                                                                      \
        const unsigned short* As = &lds[buf_][0][(wr * 64 + l31) * GM_LD + hi * 8];                        \
        const unsigned short* Bs = &lds[buf_][1][(wc * 64 + l31) * GM_LD + hi * 8];                        \
        _Pragma("unroll") for (int kk = 0; kk < GM_BK / 16; ++kk) {                                        \
            bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(As + kk * 16);                                \
            bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(As + 32 * GM_LD + kk * 16);                   \
            bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(Bs + kk * 16);                                \
            bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(Bs + 32 * GM_LD + kk * 16);                   \
            if constexpr (TR) {                                                                            \
                acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a0, acc00, 0, 0, 0);                   \
                acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a0, acc01, 0, 0, 0);                   \
                acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a1, acc10, 0, 0, 0);                   \
                acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, acc11, 0, 0, 0);                   \
            } else {                                                                                       \
                acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc00, 0, 0, 0);                   \
                acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc01, 0, 0, 0);                   \
                acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc10, 0, 0, 0);                   \
                acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc11, 0, 0, 0);                   \
            }                                                                                              \
        }                                                                                                  \
    }
    GM_LOAD_TILE(0, kt_beg);
    GM_LOAD_TILE(1, kt_beg + 1);
    GM_STORE_TILE(0, 0);
    __syncthreads();
    int kt = kt_beg;
    for (; kt + 1 < kt_end; kt += 2) {      // two tiles per trip so both register sets are addressed statically
        GM_LOAD_TILE(0, kt + 2);     // set 0 was written to LDS already; refill it two tiles ahead
        GM_COMPUTE_TILE(0);
        GM_STORE_TILE(1, 1);         // tile kt+1 (loaded one trip ago) -> the other LDS buffer
        __syncthreads();
        GM_LOAD_TILE(1, kt + 3);
        GM_COMPUTE_TILE(1);
        GM_STORE_TILE(0, 0);         // tile kt+2 (clamped duplicate past the end: harmless)
        __syncthreads();
    }
    if (kt < kt_end) GM_COMPUTE_TILE(0);    // odd tile count: the last tile sits in buffer 0
    // D^T tile layout: col (lane & 31) = output row m, row (reg&3) + 8*(reg>>2) + 4*(lane>>5) = output column n
    const bool vec_ok = (e.ldo % 4 == 0) && (e.ldaux % 4 == 0);
#define GM_EPILOGUE(acc_, i_, j_)                                                                     \
    if constexpr (TR) {                                                                               \
        const int64_t row = m0 + wr * 64 + (i_) * 32 + l31;                                           \
        if (row < M) {                                                                                \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                           \
                const int64_t c0 = n0 + wc * 64 + (j_) * 32 + 8 * g + 4 * hi;                         \
                const float v4[4] = {acc_[4 * g], acc_[4 * g + 1], acc_[4 * g + 2], acc_[4 * g + 3]}; \
                if (c0 < N) epilogue_store4<EPI, bf16>(e, row, c0, N, v4, vec_ok);                    \
            }                                                                                         \
        }                                                                                             \
    } else {                                                                                          \
        const int64_t c = n0 + wc * 64 + (j_) * 32 + l31;                                             \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                              \
            const int64_t row = m0 + wr * 64 + (i_) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;           \
            if (row < M && c < N) epilogue_store<EPI, bf16>(e, row, c, acc_[r]);                      \
        }                                                                                             \
    }
    GM_EPILOGUE(acc00, 0, 0);
    GM_EPILOGUE(acc01, 0, 1);
    GM_EPILOGUE(acc10, 1, 0);
    GM_EPILOGUE(acc11, 1, 1);
}

// ------------------------------------------------------------------------------------------------
// MFMA bf16 kernel, direct global->LDS staging (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass.
// The LDS image of a 128x64 operand tile is unpadded (128-B rows) with the 16-B chunk index XOR-ed by (row>>1)&7
// (conflict-free ds_read_b128 fragment reads); because an LDS-DMA writes wave-uniform base + lane*16, the swizzle is
// applied on the per-lane SOURCE address: lane l of the wave that fills rows 8q..8q+7 loads chunk (l&7)^swz(row) of
// row 8q + (l>>3) -- still one full 128-B line per 8 lanes.
//   NBUF == 1: 32 KB LDS, <=128 VGPRs -> 4 workgroups per CU, latency hidden by the other workgroups
//   NBUF == 2: 64 KB LDS, tile t+1 lands while tile t is multiplied, one barrier per K tile
// ------------------------------------------------------------------------------------------------

template <int EPI, int NBUF, bool GN = false>        // GN: + GroupNorm statistics of the stored output (maed_conv1x1_fwd)
__global__ __launch_bounds__(256, (NBUF == 1 ? 4 : 2)) void gemm_nt_glds_bf16_kernel(const bf16* __restrict__ A, int64_t lda,
                                                                                     const bf16* __restrict__ B, int64_t ldb, int64_t M,
                                                                                     int64_t N, int64_t K, int tiles_n,
                                                                                     int ktiles_per_split, EpiArgs e
#ifdef MAED_GEMM_ABLATE
                                                                                     , int ablate    // diagnostic build only (scripts/gemm_ablate.sh): 1 no stores, 2 no loads, 4 no MFMA
#endif
                                                                                     ) {
#ifndef MAED_GEMM_ABLATE
    constexpr int ablate = 0;
#endif
    // operand tiles [NBUF][A|B][128*64] bf16; re-used by the epilogue as 4 per-wave fp32 staging areas of 32 rows x 68 floats
    // (+ the 1-KB GroupNorm-statistics table of the convolution epilogue behind both)
    constexpr int kTileElems = NBUF * 2 * GM_BM * GM_BK, kStageElems = 4 * 32 * GL_ST * 2;     // in 2-byte units
    constexpr int kMainElems = kTileElems > kStageElems ? kTileElems : kStageElems;
    __shared__ __attribute__((aligned(1024))) unsigned short lds_raw[kMainElems + (GN ? GN_TAB_FLOATS * 2 : 0)];
    unsigned short (*lds)[2][GM_BM * GM_BK] = reinterpret_cast<unsigned short (*)[2][GM_BM * GM_BK]>(lds_raw);
    double* const gn_tab = reinterpret_cast<double*>(lds_raw + kMainElems);
    if (GN && threadIdx.x < GN_TAB_FLOATS / 2) gn_tab[threadIdx.x] = 0.0;                       // (published by the main loop's barriers)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: LDS-DMA destinations (M0) stay on the SALU
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int nwg = gridDim.x;
    const int id = xcd_remap(blockIdx.x, nwg);
    const int64_t m0 = (int64_t)(id / tiles_n) * GM_BM, n0 = (int64_t)(id % tiles_n) * GM_BN;
    const int nkt_total = (int)(K / GM_BK);
    const int kt_beg = blockIdx.z * ktiles_per_split;
    int kt_end = kt_beg + ktiles_per_split;
    if (kt_end > nkt_total) kt_end = nkt_total;
    if (kt_beg >= kt_end) return;

    // staging: round i = 0..3, this wave fills rows (4i + wave)*8 .. +7; (row>>1)&7 = (4*wave + (lane>>4)) & 7 for every round.
    // LDS-DMA in its cheapest form (MAED_LDS_DMA16): scalar base (operand + K offset) + 32-bit lane BYTE offset (host-checked < 4 GB),
    // LDS destination from a scalar wave id -- no VALU and no v_readfirstlane per DMA (8 DMAs per 16 MFMAs in this kernel).
    const int srow = wave * 8 + (lane >> 3);                                        // + 32*i
    const int schunk = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);
#define GL_PTRS(i)                                                                              \
    uint32_t gao##i, gbo##i;                                                                    \
    {                                                                                           \
        const int row = srow + 32 * i;                                                          \
        const int64_t ar = (m0 + row < M) ? m0 + row : M - 1;                                   \
        const int64_t br = (n0 + row < N) ? n0 + row : N - 1;                                   \
        gao##i = (uint32_t)((ar * lda + schunk * 8) * 2); gbo##i = (uint32_t)((br * ldb + schunk * 8) * 2); \
    }
    GL_PTRS(0) GL_PTRS(1) GL_PTRS(2) GL_PTRS(3)
    const char* const Ab = reinterpret_cast<const char*>(A);
    const char* const Bb = reinterpret_cast<const char*>(B);
#define GL_ISSUE1(i, buf_, ak__, bk__)                                          \
    MAED_LDS_DMA16(ak__, gao##i, &lds[buf_][0][(4 * i + wave) * 8 * GM_BK]);    \
    MAED_LDS_DMA16(bk__, gbo##i, &lds[buf_][1][(4 * i + wave) * 8 * GM_BK]);
#define GL_ISSUE_TILE(buf_, kt_) if (!(ablate & 2)) { const char* const ak__ = Ab + (int64_t)(kt_) * (GM_BK * 2); const char* const bk__ = Bb + (int64_t)(kt_) * (GM_BK * 2); \
        GL_ISSUE1(0, buf_, ak__, bk__) GL_ISSUE1(1, buf_, ak__, bk__) GL_ISSUE1(2, buf_, ak__, bk__) GL_ISSUE1(3, buf_, ak__, bk__) }

    constexpr bool TR = (EPI != MAED_EPI_ATOMIC_F32);
    f32x16_t acc00, acc01, acc10, acc11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }
    const int fsw = (l31 >> 1) & 7;                                                 // swizzle term of this lane's fragment rows
#define GL_COMPUTE_TILE(buf_)                                                                              \
    if (!(ablate & 4)) {                                                                                   \
        const unsigned short* As = &lds[buf_][0][(wr * 64 + l31) * GM_BK];                                 \
        const unsigned short* Bs = &lds[buf_][1][(wc * 64 + l31) * GM_BK];                                 \
        _Pragma("unroll") for (int kk = 0; kk < GM_BK / 16; ++kk) {                                        \
            const int co = ((kk * 2 + hi) ^ fsw) * 8;                                                      \
            bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(As + co);                                     \
            bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(As + 32 * GM_BK + co);                        \
            bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(Bs + co);                                     \
            bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(Bs + 32 * GM_BK + co);                        \
            if constexpr (TR) {                                                                            \
                acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a0, acc00, 0, 0, 0);                   \
                acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a0, acc01, 0, 0, 0);                   \
                acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a1, acc10, 0, 0, 0);                   \
                acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, acc11, 0, 0, 0);                   \
            } else {                                                                                       \
                acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc00, 0, 0, 0);                   \
                acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc01, 0, 0, 0);                   \
                acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc10, 0, 0, 0);                   \
                acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc11, 0, 0, 0);                   \
            }                                                                                              \
        }                                                                                                  \
    }
    if constexpr (NBUF == 1) {
        for (int kt = kt_beg; kt < kt_end; ++kt) {
            GL_ISSUE_TILE(0, kt);
            MAED_WAIT_VMCNT0();
            __syncthreads();
            GL_COMPUTE_TILE(0);
            __syncthreads();
        }
    } else {
        GL_ISSUE_TILE(0, kt_beg);
        for (int kt = kt_beg; kt < kt_end; kt += 2) {      // two tiles per trip: the buffer index is static
            MAED_WAIT_VMCNT0();
            __syncthreads();                                // tile kt landed for every wave; buffer 1 is free again
            if (kt + 1 < kt_end) GL_ISSUE_TILE(NBUF - 1, kt + 1);
            GL_COMPUTE_TILE(0);
            if (kt + 1 < kt_end) {
                MAED_WAIT_VMCNT0();
                __syncthreads();
                if (kt + 2 < kt_end) GL_ISSUE_TILE(0, kt + 2);
                GL_COMPUTE_TILE(NBUF - 1);
            }
        }
    }
    if constexpr (!TR) {
        const bool vec_ok = (e.ldo % 4 == 0) && (e.ldaux % 4 == 0);
        GM_EPILOGUE(acc00, 0, 0);
        GM_EPILOGUE(acc01, 0, 1);
        GM_EPILOGUE(acc10, 1, 0);
        GM_EPILOGUE(acc11, 1, 1);
    } else {
        // LDS-shuffled epilogue: a lane owns one output row in the accumulators (4 columns per register group), which would
        // mean 8-byte global accesses scattered over 32 rows per instruction.  Each wave parks its 32 x 64 half-tile in LDS
        // (fp32) and re-reads it so that 8 lanes cover one row's 64 columns: 16/32-byte accesses, full lines per row, for the
        // stores AND for the auxiliary reads of the residual / GELU' epilogues.
        const bool vec_ok = (e.ldo % 8 == 0) && (e.ldaux % 8 == 0);
        float* stg = reinterpret_cast<float*>(lds_raw) + wave * 32 * GL_ST;
        const int rr = lane >> 3, cc = (lane & 7) * 8;
        const GnTile gnt = GN ? gn_tile(gn_tab, m0, n0, N, e.gn_hw) : GnTile{nullptr, 0, 0, 0};
        GnRegs gnr;
        if constexpr (GN) gn_zero(gnr);
#define GL_SHUFFLE_HALF(accA_, accB_, i_)                                                                              \
        __syncthreads();                                                                                               \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                \
            *reinterpret_cast<float4*>(stg + l31 * GL_ST + 8 * g + 4 * hi) = make_float4(accA_[4 * g], accA_[4 * g + 1], accA_[4 * g + 2], accA_[4 * g + 3]);      \
            *reinterpret_cast<float4*>(stg + l31 * GL_ST + 32 + 8 * g + 4 * hi) = make_float4(accB_[4 * g], accB_[4 * g + 1], accB_[4 * g + 2], accB_[4 * g + 3]); \
        }                                                                                                              \
        __syncthreads();                                                                                               \
        _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                                             \
            const int lr = ps * 8 + rr;                                                                                \
            const int64_t row = m0 + wr * 64 + (i_) * 32 + lr, c0 = n0 + wc * 64 + cc;                                 \
            float v8[8];                                                                                               \
            ld8(stg + lr * GL_ST + cc, v8);                                                                            \
            if (row < M && c0 < N && !(ablate & 1)) epilogue_store8<EPI, bf16>(e, row, c0, N, v8, vec_ok);             \
            if constexpr (GN) { if (row < M && c0 < N) gn_acc8(gnr, gnt, v8, row); }                                   \
        }
        GL_SHUFFLE_HALF(acc00, acc01, 0)
        GL_SHUFFLE_HALF(acc10, acc11, 1)
#undef GL_SHUFFLE_HALF
        if constexpr (GN) {
            gn_commit(gnr, gnt, lane, n0 + wc * 64 + cc, N);
            __syncthreads();
            gn_flush(gnt, e.gn_sums, m0, M, e.gn_hw, GM_BN, tid, 256);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 3x3 convolution as an implicit GEMM on the same tile / LDS image / epilogue as gemm_nt_glds_bf16_kernel<EPI, 1>
// (resnetv2.py:74-93 StdConv2dSame 3x3: 16 of the backbone's 53 convolutions; today they run on MIOpen).
//   out[m][co] = sum_{tap, ci} X[pixel(m) shifted by tap][ci] * Wt[co][tap*Cin + ci]      m = (f, oy, ox), channels_last
// A-operand rows are GATHERED: every lane of the LDS-DMA computes its own source address, so "im2col" costs nothing -- for K tile
// kt the tap is (kt*64)/Cin (Cin % 64 == 0: a K tile never straddles taps), and a lane whose shifted pixel falls outside the image
// points at a 128-byte page of zeros instead (TF-SAME zero padding, any stride).  The input gradient of a stride-1 convolution is
// the same kernel on dY with the flipped, transposed weight image.  Default path of the backbone since it was timed against MIOpen
// on MI355X (resnetv2.py; profiles/r02_call1_conv3x3_micro.txt).
// ------------------------------------------------------------------------------------------------
// B (weight) addressing: element (n, tap, c) of the GEMM's B operand lives at Wt[b_base + tap * b_tap + n * b_row + c]
struct Conv3x3Dims { int F, H, W, Cin, Ho, Wo, stride, pad_top, pad_left; int64_t b_row, b_tap, b_base;
                     // CLS (one parity class of a stride-2 input gradient, maed_conv3x3_s2_dgrad): nty x ntx taps; loop tap (ty, tx) pairs with the FORWARD tap
                     // (ky0 - 2 ty, kx0 - 2 tx); output pixel (f, a, b) of the class is row (f * o_h + 2 a + o_py) * o_w + 2 b + o_px of dX
                     int nty = 3, ntx = 3, ky0 = 0, kx0 = 0, o_h = 0, o_w = 0, o_py = 0, o_px = 0; };

// CLS: row m = (f, a, b) of a parity class -> its row of dX
__device__ __forceinline__ int64_t cv_class_row(const Conv3x3Dims& d, int64_t m) {
    const int b = (int)(m % d.Wo), a = (int)((m / d.Wo) % d.Ho);
    const int64_t f = m / ((int64_t)d.Wo * d.Ho);
    return (f * d.o_h + 2 * a + d.o_py) * (int64_t)d.o_w + 2 * b + d.o_px;
}

// NARROW: 128 x 64 output tile for Cout <= 64 (stage 1 of the R50: a 128-wide tile would spend half its MFMAs on duplicated weight rows):
// the four waves take 32 pixel rows each and both 32-column halves; only 64 weight rows are staged.
template <int EPI, bool NARROW, bool GN, bool CLS = false>
__global__ __launch_bounds__(256, 4) void conv3x3_glds_bf16_kernel(const bf16* __restrict__ X, const bf16* __restrict__ Wt, const bf16* __restrict__ zero_page,
                                                                   Conv3x3Dims d, int64_t M, int64_t N, int tiles_n, EpiArgs e) {
    constexpr int kTileElems = 2 * GM_BM * GM_BK, kStageElems = 4 * 32 * GL_ST * 2;
    constexpr int kMainElems = kTileElems > kStageElems ? kTileElems : kStageElems;
    __shared__ __attribute__((aligned(1024))) unsigned short lds_raw[kMainElems + (GN ? GN_TAB_FLOATS * 2 : 0)];   // (+ GroupNorm-statistics table)
    unsigned short (*lds)[GM_BM * GM_BK] = reinterpret_cast<unsigned short (*)[GM_BM * GM_BK]>(lds_raw);     // [A|B][128*64]
    double* const gn_tab = reinterpret_cast<double*>(lds_raw + kMainElems);
    if (GN && threadIdx.x < GN_TAB_FLOATS / 2) gn_tab[threadIdx.x] = 0.0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = NARROW ? wave : wave >> 1, wc = NARROW ? 0 : wave & 1, l31 = lane & 31, hi = lane >> 5;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(id / tiles_n) * GM_BM, n0 = (int64_t)(id % tiles_n) * (NARROW ? 64 : GM_BN);
    const int nkt = (CLS ? d.nty * d.ntx : 9) * d.Cin / GM_BK;
    const int srow = wave * 8 + (lane >> 3);
    const int schunk = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);
    // per staging round i: the output pixel of this lane's A row (top-left input tap, element offset of it) and its B row
#define CV_PTRS(i)                                                                                        \
    int iy##i, ix##i; int64_t aoff##i; const bf16* gbp##i;                                                \
    {                                                                                                     \
        const int row = srow + 32 * i;                                                                    \
        const int64_t m = (m0 + row < M) ? m0 + row : M - 1;                                              \
        const int ox = (int)(m % d.Wo), oy = (int)((m / d.Wo) % d.Ho);                                    \
        const int64_t f = m / ((int64_t)d.Wo * d.Ho);                                                     \
        iy##i = oy * d.stride - d.pad_top; ix##i = ox * d.stride - d.pad_left;                            \
        aoff##i = ((f * d.H + iy##i) * d.W + ix##i) * (int64_t)d.Cin + schunk * 8;                        \
        const int64_t br = (n0 + row < N) ? n0 + row : N - 1;                                             \
        gbp##i = Wt + d.b_base + br * d.b_row + schunk * 8;                                               \
    }
    CV_PTRS(0) CV_PTRS(1) CV_PTRS(2) CV_PTRS(3)
#define CV_ISSUE1(i, ty_, tx_, toff_, k0_)                                                                                            \
    {                                                                                                                                 \
        const bool ok = (unsigned)(iy##i + ty_) < (unsigned)d.H && (unsigned)(ix##i + tx_) < (unsigned)d.W;                           \
        const bf16* src = ok ? X + aoff##i + toff_ : zero_page + schunk * 8;                                                          \
        __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)&lds[0][(4 * i + wave) * 8 * GM_BK], 16, 0, 0);               \
        if (!NARROW || i < 2)                                                                                                         \
            __builtin_amdgcn_global_load_lds((glb_void_t*)(gbp##i + k0_), (lds_void_t*)&lds[1][(4 * i + wave) * 8 * GM_BK], 16, 0, 0); \
    }
    f32x16_t acc00, acc01, acc10, acc11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }
    const int fsw = (l31 >> 1) & 7;
    int ty = 0, tx = 0, c0 = 0;                                     // K tile -> (tap, channel chunk), advanced incrementally (wave-uniform)
    for (int kt = 0; kt < nkt; ++kt) {
        const int64_t k0 = (int64_t)(CLS ? (d.ky0 - 2 * ty) * 3 + (d.kx0 - 2 * tx) : ty * 3 + tx) * d.b_tap + c0;     // B offset of this K tile
        const int64_t toff = ((int64_t)ty * d.W + tx) * d.Cin + c0;
        CV_ISSUE1(0, ty, tx, toff, k0) CV_ISSUE1(1, ty, tx, toff, k0) CV_ISSUE1(2, ty, tx, toff, k0) CV_ISSUE1(3, ty, tx, toff, k0)
        MAED_WAIT_VMCNT0();
        __syncthreads();
        {
            const unsigned short* As = &lds[0][(wr * (NARROW ? 32 : 64) + l31) * GM_BK];
            const unsigned short* Bs = &lds[1][(wc * 64 + l31) * GM_BK];
#pragma unroll
            for (int kk = 0; kk < GM_BK / 16; ++kk) {
                const int co = ((kk * 2 + hi) ^ fsw) * 8;
                const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(As + co);
                const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(Bs + co);
                const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(Bs + 32 * GM_BK + co);
                acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a0, acc00, 0, 0, 0);     // transposed tiles: lane = output row (pixel)
                acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a0, acc01, 0, 0, 0);
                if constexpr (!NARROW) {
                    const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(As + 32 * GM_BK + co);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a1, acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, acc11, 0, 0, 0);
                }
            }
        }
        __syncthreads();
        c0 += GM_BK;
        if (c0 == d.Cin) { c0 = 0; if (++tx == (CLS ? d.ntx : 3)) { tx = 0; ++ty; } }
    }
#undef CV_PTRS
#undef CV_ISSUE1
    // LDS-shuffled epilogue, exactly as in gemm_nt_glds_bf16_kernel
    const bool vec_ok = (e.ldo % 8 == 0) && (e.ldaux % 8 == 0);
    float* stg = reinterpret_cast<float*>(lds_raw) + wave * 32 * GL_ST;
    const int rr = lane >> 3, cc = (lane & 7) * 8;
    const GnTile gnt = GN ? gn_tile(gn_tab, m0, n0, N, e.gn_hw) : GnTile{nullptr, 0, 0, 0};
    GnRegs gnr;
    if constexpr (GN) gn_zero(gnr);
#define CV_SHUFFLE_HALF(accA_, accB_, i_)                                                                              \
    __syncthreads();                                                                                                   \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                    \
        *reinterpret_cast<float4*>(stg + l31 * GL_ST + 8 * g + 4 * hi) = make_float4(accA_[4 * g], accA_[4 * g + 1], accA_[4 * g + 2], accA_[4 * g + 3]);      \
        *reinterpret_cast<float4*>(stg + l31 * GL_ST + 32 + 8 * g + 4 * hi) = make_float4(accB_[4 * g], accB_[4 * g + 1], accB_[4 * g + 2], accB_[4 * g + 3]); \
    }                                                                                                                  \
    __syncthreads();                                                                                                   \
    _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                                                 \
        const int lr = ps * 8 + rr;                                                                                    \
        const int64_t row = m0 + wr * (NARROW ? 32 : 64) + (i_) * 32 + lr, c0 = n0 + wc * 64 + cc;                     \
        float v8[8];                                                                                                   \
        ld8(stg + lr * GL_ST + cc, v8);                                                                                \
        if (row < M && c0 < N) epilogue_store8<EPI, bf16>(e, CLS ? cv_class_row(d, row) : row, c0, N, v8, vec_ok);     \
        if constexpr (GN) { if (row < M && c0 < N) gn_acc8(gnr, gnt, v8, row); }                                       \
    }
    CV_SHUFFLE_HALF(acc00, acc01, 0)
    if constexpr (!NARROW) { CV_SHUFFLE_HALF(acc10, acc11, 1) }
#undef CV_SHUFFLE_HALF
    if constexpr (GN) {
        gn_commit(gnr, gnt, lane, n0 + wc * 64 + cc, N);
        __syncthreads();
        gn_flush(gnt, e.gn_sums, m0, M, e.gn_hw, NARROW ? 64 : GM_BN, tid, 256);
    }
}


// ------------------------------------------------------------------------------------------------
// Round 6 (second session): the same implicit GEMM on ONE FRAME x 128 output channels per workgroup, for feature maps of at most 256 pixels (stage 3 of the R50:
// 14 x 14 = 196; cfg5: 16 x 16).  Why: at stage 3 the 128 x 128 tiling gives 392 workgroups -- 1.5 per CU.  Its single-buffered loop costs a workgroup one exposed copy
// round trip per K tile (36 of them: ~45 us whoever shares the CU), and the CUs that got two workgroups move 2.3 MB through the LDS-DMA path in that time, which is what
// that path gives a CU (~50 GB/s): latency-bound and fill-bound at once, more workgroups of the same kind re-read more (r06_conv3x3_narrow_tiles_rejected.txt).  A frame
// tile has NO imbalance (128 frames x 2 column tiles = 256 workgroups = one per CU), moves 1.6 MB per CU (a frame's 196 rows + 128 weight rows per K tile: the weight
// rows are shared by twice the pixels) and, alone on its CU, can afford a ring: three stages of 48 KB, copies two K tiles ahead, ONE barrier per K tile.
// Eight waves: wave w owns pixel rows 32 w .. 32 w + 31 of the frame (rows past the frame: nothing to copy, nothing to multiply) against all 128 columns
// (four 32 x 32 accumulators); GroupNorm statistics as in the 128-row kernel (a tile is one frame: the table's second frame stays empty).
// ------------------------------------------------------------------------------------------------
#define CF_STAGES 3
#define CF_A_ELEMS (256 * GM_BK)
#define CF_STAGE_ELEMS (CF_A_ELEMS + 128 * GM_BK)          // 48 KB
template <int EPI, bool GN>
__global__ __launch_bounds__(512, 2) void conv3x3_frame_bf16_kernel(const bf16* __restrict__ X, const bf16* __restrict__ Wt, const bf16* __restrict__ zero_page,
                                                                    Conv3x3Dims d, int64_t M, int64_t N, int tiles_n, EpiArgs e) {
    MAED_DYN_SHARED(unsigned short, lds);                     // CF_STAGES x [A: 256 x 64][B: 128 x 64] (+ the GroupNorm-statistics table)
    double* const gn_tab = reinterpret_cast<double*>(lds + CF_STAGES * CF_STAGE_ELEMS);
    if (GN && threadIdx.x < GN_TAB_FLOATS / 2) gn_tab[threadIdx.x] = 0.0;          // (published by the main loop's barriers)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int HW = d.Ho * d.Wo;
    const int f = id / tiles_n;
    const int64_t m0 = (int64_t)f * HW, n0 = (int64_t)(id % tiles_n) * 128;
    const bool active = wave * 32 < HW;                       // wave-uniform: this wave has pixel rows
    const int nkt = 9 * d.Cin / GM_BK;
    // copies: round j of A = this wave's rows 8 j .. 8 j + 7 (one 1 KB instruction), round j of B = weight rows 16 wave + 8 j ..; the 16-byte chunk a lane copies
    // undoes the fragment reads' swizzle: slot (lane & 7) of row r holds chunk slot ^ ((r >> 1) & 7), and (r >> 1) & 7 = (4 j + (lane >> 4)) & 7 for both operands
    int iy[4], ix[4]; int64_t aoff[4]; const bf16* gbp[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int sch = (lane & 7) ^ ((4 * j + (lane >> 4)) & 7);
        int p = wave * 32 + 8 * j + (lane >> 3);
        if (p > HW - 1) p = HW - 1;
        const int ox = p % d.Wo, oy = p / d.Wo;
        iy[j] = oy * d.stride - d.pad_top; ix[j] = ox * d.stride - d.pad_left;
        aoff[j] = (((int64_t)f * d.H + iy[j]) * d.W + ix[j]) * (int64_t)d.Cin + sch * 8;
        if (j < 2) {
            const int64_t br = n0 + 16 * wave + 8 * j + (lane >> 3);
            gbp[j] = Wt + d.b_base + (br < N ? br : N - 1) * d.b_row + sch * 8;
        }
    }
    const int zch = ((lane & 7)) * 8;                         // any chunk of the zero page
#define CF_ISSUE(stage_, ty_, tx_, toff_, k0_) {                                                                                          \
        unsigned short* const sa__ = lds + (stage_) * CF_STAGE_ELEMS;                                                                     \
        if (active) {                                                                                                                     \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                               \
                const bool ok = (unsigned)(iy[j] + (ty_)) < (unsigned)d.H && (unsigned)(ix[j] + (tx_)) < (unsigned)d.W;                   \
                const bf16* src = ok ? X + aoff[j] + (toff_) : zero_page + zch;                                                           \
                MAED_LDS_DMA16_PTR(src, sa__ + (32 * wave + 8 * j) * GM_BK);                                                              \
            }                                                                                                                             \
        }                                                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                                     \
            MAED_LDS_DMA16_PTR(gbp[j] + (k0_), sa__ + CF_A_ELEMS + (16 * wave + 8 * j) * GM_BK);                                          \
    }
    f32x16_t acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    const int fsw = (l31 >> 1) & 7;
    // the copy stream runs two K tiles ahead of the products: its own (tap, channel chunk) counters
    int ity = 0, itx = 0, ic0 = 0;
#define CF_ISSUE_NEXT(stage_) {                                                                                                           \
        const int64_t k0__ = (int64_t)(ity * 3 + itx) * d.b_tap + ic0;                                                                    \
        const int64_t toff__ = ((int64_t)ity * d.W + itx) * d.Cin + ic0;                                                                  \
        CF_ISSUE(stage_, ity, itx, toff__, k0__)                                                                                          \
        ic0 += GM_BK;                                                                                                                     \
        if (ic0 == d.Cin) { ic0 = 0; if (++itx == 3) { itx = 0; ++ity; } }                                                                \
    }
    CF_ISSUE_NEXT(0)
    CF_ISSUE_NEXT(1)                                            // (nkt >= 9)
    for (int kt = 0; kt < nkt; ++kt) {
        // tile kt has landed once at most the younger tile's copies (6 per active wave, 2 per idle one) are outstanding
        if (kt + 1 < nkt) { if (active) { MAED_WAIT_VMCNT(6); } else { MAED_WAIT_VMCNT(2); } } else { MAED_WAIT_VMCNT0(); }
        __syncthreads();                                        // ... for every wave; and every wave is done with tile kt - 1: its stage is free
        if (kt + 2 < nkt) CF_ISSUE_NEXT((kt + 2) % CF_STAGES)
        if (active) {
            const unsigned short* const st = lds + (kt % CF_STAGES) * CF_STAGE_ELEMS;
            const unsigned short* As = st + (wave * 32 + l31) * GM_BK;
            const unsigned short* Bs = st + CF_A_ELEMS + l31 * GM_BK;
#pragma unroll
            for (int kk = 0; kk < GM_BK / 16; ++kk) {
                const int co = ((kk * 2 + hi) ^ fsw) * 8;
                const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(As + co);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(Bs + nb * 32 * GM_BK + co);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a0, acc[nb], 0, 0, 0);      // transposed tiles: lane = output row (pixel)
                }
            }
        }
    }
#undef CF_ISSUE_NEXT
#undef CF_ISSUE
    // LDS-shuffled epilogue as in the 128-row kernels: a wave's 32 x 64 half through its private staging area, rows out as 8-column pieces
    const bool vec_ok = (e.ldo % 8 == 0) && (e.ldaux % 8 == 0);
    float* stg = reinterpret_cast<float*>(lds) + wave * 32 * GL_ST;
    const int rr = lane >> 3, cc = (lane & 7) * 8;
    const GnTile gnt = GN ? gn_tile(gn_tab, m0, n0, N, HW) : GnTile{nullptr, 0, 0, 0};
    GnRegs gnr;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();                                        // the ring (first pass) / this wave's staging rows (second pass) are no longer read
        if constexpr (GN) gn_zero(gnr);
        if (active) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<float4*>(stg + l31 * GL_ST + 8 * g + 4 * hi) = make_float4(acc[2 * h][4 * g], acc[2 * h][4 * g + 1], acc[2 * h][4 * g + 2], acc[2 * h][4 * g + 3]);
                *reinterpret_cast<float4*>(stg + l31 * GL_ST + 32 + 8 * g + 4 * hi) = make_float4(acc[2 * h + 1][4 * g], acc[2 * h + 1][4 * g + 1], acc[2 * h + 1][4 * g + 2], acc[2 * h + 1][4 * g + 3]);
            }
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int lr = ps * 8 + rr;
                const int pr = wave * 32 + lr;
                const int64_t row = m0 + pr, c0 = n0 + h * 64 + cc;
                float v8[8];
                ld8(stg + lr * GL_ST + cc, v8);
                if (pr < HW && c0 < N) epilogue_store8<EPI, bf16>(e, row, c0, N, v8, vec_ok);
                if constexpr (GN) { if (pr < HW && c0 < N) gn_acc8(gnr, gnt, v8, row); }
            }
            if constexpr (GN) gn_commit(gnr, gnt, lane, n0 + h * 64 + cc, N);
        }
    }
    if constexpr (GN) {
        __syncthreads();
        gn_flush(gnt, e.gn_sums, m0, M, HW, 128, tid, 512);
    }
}

static bool gn_stats_shape_ok(int64_t channels, int64_t hw) {          // 32 groups of 2^k channels; a 128-row tile spans at most two frames
    const int64_t cpg = channels / 32;
    return channels % 32 == 0 && cpg >= 2 && (cpg & (cpg - 1)) == 0 && hw >= 128;
}

extern "C" int maed_conv3x3_fwd(const void* x, const void* w_taps, const void* zero_page, void* y, int F, int H, int W, int Cin, int Cout,
                                int stride, int pad_top, int pad_left, int Ho, int Wo, const void* add, int w_layout, int dtype, double* gn_sums,
                                void* stream) {
    MAED_CHECK_ARG(!gn_sums || (gn_stats_shape_ok(Cout, (int64_t)Ho * Wo) && !add), MAED_ERR_SHAPE,
                   "conv3x3_fwd: GroupNorm statistics need Cout = 32 * 2^k >= 64, Ho*Wo >= 128 and no `add` (Cout=%d Ho*Wo=%d)", Cout, Ho * Wo);
    MAED_CHECK_ARG(w_layout == 0 || w_layout == 1, MAED_ERR_ARG, "conv3x3_fwd: w_layout must be 0 (Cout,3,3,Cin) or 1 (transposed image of the forward weight)");
    MAED_CHECK_ARG(x && w_taps && zero_page && y, MAED_ERR_ARG, "conv3x3_fwd: null pointer");
    const int np_call = maed_x3_take_dtype(dtype);
    MAED_CHECK_ARG(dtype == MAED_BF16 || dtype == MAED_F32, MAED_ERR_ARG, "conv3x3_fwd: bad dtype %d", dtype);
    const int x3np = dtype == MAED_F32 ? (np_call ? np_call : maed_x3_planes()) : 0;
    MAED_CHECK_ARG(dtype == MAED_BF16 || x3np, MAED_ERR_UNSUPPORTED, "conv3x3_fwd: f32 needs the split-bf16 matmul mode (maed_set_option(MAED_OPT_F32_MATMUL, 1 or 2))");
    MAED_CHECK_ARG(F >= 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && stride >= 1 && pad_top >= 0 && pad_left >= 0, MAED_ERR_SHAPE, "conv3x3_fwd: bad extents");
    MAED_CHECK_ARG(Cin % (dtype == MAED_F32 ? 32 : GM_BK) == 0 && Cout % 8 == 0, MAED_ERR_SHAPE, "conv3x3_fwd: need Cin %% 64 == 0 (f32: 32) and Cout %% 8 == 0 (Cin=%d Cout=%d)", Cin, Cout);
    MAED_CHECK_ARG((Ho - 1) * stride - pad_top + 2 < H + 2 && (Wo - 1) * stride - pad_left + 2 < W + 2, MAED_ERR_SHAPE, "conv3x3_fwd: output extent exceeds the padded input");
    MAED_CHECK_ARG(is_aligned(x, 16) && is_aligned(w_taps, 16) && is_aligned(zero_page, 16) && is_aligned(y, 16), MAED_ERR_ALIGN, "conv3x3_fwd: 16-B alignment");
    if (F == 0) return MAED_OK;
    const int64_t M = (int64_t)F * Ho * Wo, N = Cout;
    const int tm = (int)((M + GM_BM - 1) / GM_BM);
    // 128 x 64 output tiles for Cout <= 64, and (MAED_OPT_CONV3X3_NARROW_WGS) wherever 128 x 128 tiles would leave the chip with too few workgroups to hide the
    // single-buffered loop's copy latency (stage 3 of the R50: 392 workgroups = 1.5 per CU)
    const bool narrow = N <= 64 || (int64_t)tm * ((N + GM_BN - 1) / GM_BN) < maed_opt(MAED_OPT_CONV3X3_NARROW_WGS);
    const int tn = narrow ? (int)((N + 63) / 64) : (int)((N + GM_BN - 1) / GM_BN);
    // layout 0: w_taps[co][tap][ci].  layout 1 (input gradient from the forward weight's transposed image Wt[tap_f][c_f][o_f], as
    // maed_weight_std_fwd writes it next to the forward image): here Cin = O_f, Cout = I_f, and element (n = c_f, tap, c = o_f) is
    // Wt[(8 - tap)][n][c] -- the tap flip is a negative tap stride, nothing is copied.
    const Conv3x3Dims d = w_layout == 0
        ? Conv3x3Dims{F, H, W, Cin, Ho, Wo, stride, pad_top, pad_left, 9 * (int64_t)Cin, (int64_t)Cin, 0}
        : Conv3x3Dims{F, H, W, Cin, Ho, Wo, stride, pad_top, pad_left, (int64_t)Cin, -(int64_t)Cout * Cin, 8 * (int64_t)Cout * Cin};
    EpiArgs e{nullptr, y, (int64_t)Cout, nullptr, add, (int64_t)Cout, gn_sums, Ho * Wo};
    if (dtype == MAED_F32) {        // fp32 operands on the split-bf16 MFMA kernel (gemm_x3.hip): out-of-image taps are zeros in registers, zero_page unused
        const X3ConvDims xd{d.F, d.H, d.W, d.Cin, d.Ho, d.Wo, d.stride, d.pad_top, d.pad_left, d.b_row, d.b_tap, d.b_base};
        MAED_PROPAGATE(maed_conv3x3_x3_launch(x3np, x, w_taps, xd, M, Cout, e, add != nullptr, gn_sums != nullptr, (hipStream_t)stream));
        MAED_CHECK_LAUNCH("conv3x3_fwd(x3)");
        return MAED_OK;
    }
    // one frame x 128 channels per workgroup where a frame is at most 256 pixels and the frames fill the chip (stage 3 of the R50)
    const int fhw = Ho * Wo;
    const int fopt = maed_opt(MAED_OPT_CONV3X3_FRAME);     // 2: whenever the shape allows (tests)
    if (fopt && fhw <= 256 && fhw > 128 && N % 8 == 0 && ((int64_t)F * ((N + 127) / 128) >= 192 || fopt == 2) && (!gn_sums || gn_stats_shape_ok(Cout, fhw))) {
        const int ftn = (int)((N + 127) / 128);
        constexpr size_t lds_bytes = (size_t)CF_STAGES * CF_STAGE_ELEMS * 2 + GN_TAB_FLOATS * 4;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)conv3x3_frame_bf16_kernel<MAED_EPI_ADD, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            (void)hipFuncSetAttribute((const void*)conv3x3_frame_bf16_kernel<MAED_EPI_STORE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            (void)hipFuncSetAttribute((const void*)conv3x3_frame_bf16_kernel<MAED_EPI_STORE, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            attr_set = true;
        }
#define CF_LAUNCH(EPI_, GN_) hipLaunchKernelGGL((conv3x3_frame_bf16_kernel<EPI_, GN_>), dim3((unsigned)(F * ftn)), dim3(512), lds_bytes, (hipStream_t)stream, (const bf16*)x, \
                                                (const bf16*)w_taps, (const bf16*)zero_page, d, M, N, ftn, e)
        if (add) CF_LAUNCH(MAED_EPI_ADD, false); else if (gn_sums) CF_LAUNCH(MAED_EPI_STORE, true); else CF_LAUNCH(MAED_EPI_STORE, false);
#undef CF_LAUNCH
        MAED_CHECK_LAUNCH("conv3x3_fwd(frame)");
        return MAED_OK;
    }
    const dim3 grid((unsigned)(tm * tn));
#define CV_LAUNCH(EPI_, NARROW_, GN_) hipLaunchKernelGGL((conv3x3_glds_bf16_kernel<EPI_, NARROW_, GN_>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)x, \
                                                   (const bf16*)w_taps, (const bf16*)zero_page, d, M, N, tn, e)
    if (add) { if (narrow) CV_LAUNCH(MAED_EPI_ADD, true, false); else CV_LAUNCH(MAED_EPI_ADD, false, false); }
    else if (gn_sums) { if (narrow) CV_LAUNCH(MAED_EPI_STORE, true, true); else CV_LAUNCH(MAED_EPI_STORE, false, true); }
    else { if (narrow) CV_LAUNCH(MAED_EPI_STORE, true, false); else CV_LAUNCH(MAED_EPI_STORE, false, false); }
#undef CV_LAUNCH
    MAED_CHECK_LAUNCH("conv3x3_fwd");
    return MAED_OK;
}

// Input gradient of a STRIDE-2 3x3 SAME convolution (resnetv2.py:74-93: conv2 of the first block of stages 2 and 3) as four implicit GEMMs, one per parity
// class of the input pixel: dX[f, iy, ix, :] = sum over the forward taps (ky, kx) with (iy + pad_top - ky) and (ix + pad_left - kx) EVEN of
// dY[f, (iy + pad_top - ky) / 2, (ix + pad_left - kx) / 2, :] W[:, ky, kx, :] -- a pixel of parity (py, px) sees 2 or 1 taps per axis (9/4 of the dense
// kernel's multiply-adds in total), each class is a small stride-1 convolution over dY whose outputs land on every second row / column of dX.  Same kernel as the
// forward (gathered LDS-DMA rows, zero page for taps outside dY), transposed forward-weight image read in place; the four launches cover dX: no zero-fill.
// Replaces MIOpen's backward-data solver for these two layers together with the padded / sliced copies its symmetric-padding interface forced
// (0.3 ms per cfg3 step, profiles/r03_rocprofv3_last_step_kernel_sequence.txt).
extern "C" int maed_conv3x3_s2_dgrad(const void* dy, const void* wt_image, const void* zero_page, void* dx, int F, int H, int W, int Cin, int Cout,
                                     int pad_top, int pad_left, int Ho, int Wo, int dtype, void* stream) {
    // H, W, Cin: the forward convolution's INPUT (= dX) extents and channels; Ho, Wo, Cout: its output (= dY); wt_image (3,3,Cin,Cout) as maed_weight_std_fwd writes it
    MAED_CHECK_ARG(dy && wt_image && zero_page && dx, MAED_ERR_ARG, "conv3x3_s2_dgrad: null pointer");
    MAED_CHECK_ARG(dtype == MAED_BF16, MAED_ERR_UNSUPPORTED, "conv3x3_s2_dgrad: bf16 only (dtype=%d)", dtype);
    MAED_CHECK_ARG(F >= 0 && H > 1 && W > 1 && Ho > 0 && Wo > 0 && pad_top >= 0 && pad_top <= 1 && pad_left >= 0 && pad_left <= 1, MAED_ERR_SHAPE, "conv3x3_s2_dgrad: bad extents");
    MAED_CHECK_ARG(Cout % GM_BK == 0 && Cin % 8 == 0, MAED_ERR_SHAPE, "conv3x3_s2_dgrad: need Cout %% 64 == 0 and Cin %% 8 == 0 (Cin=%d Cout=%d)", Cin, Cout);
    MAED_CHECK_ARG(is_aligned(dy, 16) && is_aligned(wt_image, 16) && is_aligned(zero_page, 16) && is_aligned(dx, 16), MAED_ERR_ALIGN, "conv3x3_s2_dgrad: 16-B alignment");
    MAED_CHECK_ARG((uint64_t)F * H * W * Cin * 2 < (1ull << 32) && (uint64_t)F * Ho * Wo * Cout * 2 < (1ull << 32), MAED_ERR_SHAPE, "conv3x3_s2_dgrad: tensor larger than 4 GB");
    if (F == 0) return MAED_OK;
    const int64_t N = Cin;
    const bool narrow = N <= 64;
    const int tn = narrow ? 1 : (int)((N + GM_BN - 1) / GM_BN);
    EpiArgs e{nullptr, dx, (int64_t)Cin, nullptr, nullptr, (int64_t)Cin};
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            const int Ha = (H - py + 1) / 2, Wb = (W - px + 1) / 2;                 // pixels of this class per frame
            if (Ha <= 0 || Wb <= 0) continue;
            const int ey = (py + pad_top) & 1, ex = (px + pad_left) & 1;            // parity the forward tap must have
            const int nty = ey ? 1 : 2, ntx = ex ? 1 : 2;
            // forward taps ky = ey + 2 jy read dY row a + (py + pad_top - ey) / 2 - jy; loop tap ty = nty - 1 - jy walks the rows upwards
            const int base_y = (py + pad_top - ey) / 2 - (nty - 1), base_x = (px + pad_left - ex) / 2 - (ntx - 1);
            Conv3x3Dims d{F, Ho, Wo, Cout, Ha, Wb, 1, -base_y, -base_x, (int64_t)Cout, (int64_t)Cin * Cout, 0};
            d.nty = nty; d.ntx = ntx; d.ky0 = ey + 2 * (nty - 1); d.kx0 = ex + 2 * (ntx - 1); d.o_h = H; d.o_w = W; d.o_py = py; d.o_px = px;
            const int64_t M = (int64_t)F * Ha * Wb;
            const dim3 grid((unsigned)(((M + GM_BM - 1) / GM_BM) * tn));
            if (narrow) hipLaunchKernelGGL((conv3x3_glds_bf16_kernel<MAED_EPI_STORE, true, false, true>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)dy,
                                           (const bf16*)wt_image, (const bf16*)zero_page, d, M, N, tn, e);
            else hipLaunchKernelGGL((conv3x3_glds_bf16_kernel<MAED_EPI_STORE, false, false, true>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)dy,
                                    (const bf16*)wt_image, (const bf16*)zero_page, d, M, N, tn, e);
        }
    MAED_CHECK_LAUNCH("conv3x3_s2_dgrad");
    return MAED_OK;
}

// ------------------------------------------------------------------------------------------------
// out[r][c] = bias[c] (or 0): the starting value of a split-K accumulation
__global__ __launch_bounds__(256) void bias_fill_kernel(float* __restrict__ out, int64_t ldo, const float* __restrict__ bias, int64_t M, int64_t N) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < M * N) { const int64_t r = i / N, c = i % N; out[r * ldo + c] = bias ? bias[c] : 0.f; }
}

template <int EPI, typename T>
static int launch_valu(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K,
                       const EpiArgs& e, int splitk, hipStream_t s) {
    int64_t kps = (K + splitk - 1) / splitk;
    kps = (kps + 15) / 16 * 16;
    const int z = (int)((K + kps - 1) / kps);
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64), (unsigned)z);
    hipLaunchKernelGGL((gemm_nt_valu_kernel<EPI, T>), grid, dim3(256), 0, s, (const T*)A, lda, (const T*)B, ldb, M, N, K, kps, e);
    return MAED_OK;
}

template <int EPI>
static int launch_mfma(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K,
                       const EpiArgs& e, int splitk, hipStream_t s) {
    const int tm = (int)((M + GM_BM - 1) / GM_BM), tn = (int)((N + GM_BN - 1) / GM_BN);
    const int nkt = (int)(K / GM_BK);
    int kps = (nkt + splitk - 1) / splitk;
    if (kps < 1) kps = 1;
    const int z = (nkt + kps - 1) / kps;
    dim3 grid((unsigned)(tm * tn), 1, (unsigned)z);
    hipLaunchKernelGGL((gemm_nt_mfma_bf16_kernel<EPI>), grid, dim3(256), 0, s, (const bf16*)A, lda, (const bf16*)B, ldb,
                       M, N, K, tn, kps, e);
    return MAED_OK;
}

template <int EPI, int NBUF, bool GN = false>
static int launch_glds(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K,
                       const EpiArgs& e, int splitk, hipStream_t s) {
    const int tm = (int)((M + GM_BM - 1) / GM_BM), tn = (int)((N + GM_BN - 1) / GM_BN);
    const int nkt = (int)(K / GM_BK);
    int kps = (nkt + splitk - 1) / splitk;
    if (kps < 1) kps = 1;
    const int z = (nkt + kps - 1) / kps;
#ifdef MAED_GEMM_ABLATE
    hipLaunchKernelGGL((gemm_nt_glds_bf16_kernel<EPI, NBUF, GN>), dim3((unsigned)(tm * tn), 1, (unsigned)z), dim3(256), 0, s, (const bf16*)A, lda,
                       (const bf16*)B, ldb, M, N, K, tn, kps, e, maed_opt(MAED_OPT_ABLATE));
#else
    hipLaunchKernelGGL((gemm_nt_glds_bf16_kernel<EPI, NBUF, GN>), dim3((unsigned)(tm * tn), 1, (unsigned)z), dim3(256), 0, s, (const bf16*)A, lda,
                       (const bf16*)B, ldb, M, N, K, tn, kps, e);
#endif
    return MAED_OK;
}

// csrc/gemm256.hip: 256x256 tiles with the counted-vmcnt LDS-DMA pipeline
bool maed_gemm_nt_256_launch(int epilogue, const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const EpiArgs& e,
                             hipStream_t s);

// csrc/gemm_sk.hip: persistent K-stream kernel (256x256 tiles, one workgroup per CU, stream-K cuts)
bool maed_gemm_nt_sk_shape_ok(int64_t M, int64_t N, int64_t K);
bool maed_gemm_nt_sk_launch(int epilogue, const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const EpiArgs& e,
                            int mode, int grid_opt, hipStream_t s);
int maed_sk_cus(void);

template <int EPI>
static int dispatch(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, int dtype,
                    const EpiArgs& e, int splitk, int impl, hipStream_t s) {
    if (dtype == MAED_F32) {
        MAED_CHECK_ARG(impl == MAED_IMPL_AUTO || impl == MAED_IMPL_VALU || impl == MAED_IMPL_X3 || impl == MAED_IMPL_X6 || impl == MAED_IMPL_X1, MAED_ERR_UNSUPPORTED,
                       "gemm_nt: f32 runs on the exact-f32 VALU kernel (MAED_IMPL_VALU) or the split-bf16 MFMA kernel (MAED_IMPL_X3 / _X6); impl=%d", impl);
        // split-bf16 MFMA kernel (gemm_x3.hip): explicitly, or when the process-wide fp32 matmul mode asks for it.  GEMMs with few output tiles
        // (ts_attn, the decoder head: M = frames) take the split-K route below -- a 128-row tile per workgroup would leave most of the chip idle.
        const int np = impl == MAED_IMPL_X3 ? 2 : impl == MAED_IMPL_X6 ? 3 : impl == MAED_IMPL_X1 ? 1 : impl == MAED_IMPL_AUTO ? maed_x3_planes() : 0;
        if (np) {
            const bool ok = maed_x3_nt_shape_ok(A, lda, B, ldb, K);       // (otherwise: the exact kernel below -- never less accurate than asked for)
            const int64_t tiles128 = ((M + 127) / 128) * ((N + 127) / 128);
            if constexpr (EPI == MAED_EPI_STORE || EPI == MAED_EPI_STORE_F32) {
                // an EXPLICIT split engine on a GEMM with few output tiles and a long K (the decoder head in the bf16 mode, round 4: 128 frames x 1024 x 512 is 8 tiles
                // of 128 x 128): the split-K route of the exact kernel below, on the matrix cores -- bias fill, then K slices that meet with fp32 atomics
                // (never for a call that also leaves bf16 twins / planes: the atomic epilogue of the K slices writes the fp32 result only -- ADVICE r5)
                if (ok && impl != MAED_IMPL_AUTO && splitk == 1 && tiles128 < 48 && K >= 256 && e.ldo >= N && !e.twin && !e.lo && !e.out2_bf16) {
                    int sk = (int)(256 / tiles128);
                    if (sk > K / 32) sk = (int)(K / 32);
                    if (sk > 1) {
                        hipLaunchKernelGGL(bias_fill_kernel, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, s, (float*)e.out, e.ldo, e.bias, M, N);
                        EpiArgs ea = e;
                        ea.bias = nullptr;
                        return maed_gemm_nt_x3_launch(MAED_EPI_ATOMIC_F32, np, A, lda, B, ldb, M, N, K, ea, sk, s);
                    }
                }
            }
            if (ok && (impl != MAED_IMPL_AUTO || tiles128 >= 48 || EPI == MAED_EPI_ATOMIC_F32))
                return maed_gemm_nt_x3_launch(EPI, np, A, lda, B, ldb, M, N, K, e, splitk, s);
        }
        // few output tiles, long K (the decoder tail's GEMMs: 128 frames x 1024 x 1024 is 32 tiles of 64 x 64 -- 86 us on 32 CUs): spread
        // K over the chip -- the fp32 output starts as the bias and the K slices accumulate with fp32 atomics (15 us)
        if constexpr (EPI == MAED_EPI_STORE || EPI == MAED_EPI_STORE_F32) {
            const int64_t tiles = ((M + 63) / 64) * ((N + 63) / 64);
            if (splitk == 1 && tiles < 96 && K >= 256 && !e.twin && !e.lo && !e.out2_bf16) {
                // more K slices would hide more load latency, but every slice adds an output tile of fp32 atomics: ~256 workgroups measured best (128 x 1024 x 2136: 50.7 / 55.4 / 69.3 us
                // at 256 / 512 / 1024)
                constexpr int target = 256;
                int sk = (int)(target / tiles);
                if (sk > K / 64) sk = (int)(K / 64);
                if (sk > 1) {
                    hipLaunchKernelGGL(bias_fill_kernel, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, s, (float*)e.out, e.ldo, e.bias, M, N);
                    EpiArgs ea{nullptr, e.out, e.ldo, nullptr, nullptr, 0};
                    // split modes: the K slices on the split-bf16 MFMA kernel (128 x 128 tiles, atomic epilogue) -- 128 x 1024 x 1024: 32 -> 12 us
                    if (np && maed_x3_nt_shape_ok(A, lda, B, ldb, K)) {
                        const int64_t t128 = ((M + 127) / 128) * ((N + 127) / 128);
                        int sk3 = (int)(256 / t128);
                        if (sk3 > K / 64) sk3 = (int)(K / 64);
                        if (sk3 >= 1) return maed_gemm_nt_x3_launch(MAED_EPI_ATOMIC_F32, np, A, lda, B, ldb, M, N, K, ea, sk3, s);
                    }
                    return launch_valu<MAED_EPI_ATOMIC_F32, float>(A, lda, B, ldb, M, N, K, ea, sk, s);
                }
            }
        }
        return launch_valu<EPI, float>(A, lda, B, ldb, M, N, K, e, splitk, s);
    }
    // (the direct-to-LDS kernels address the operands with 32-bit lane byte offsets from a scalar base)
    const bool fits32 = (uint64_t)M * (uint64_t)lda * 2 < (1ull << 32) && (uint64_t)N * (uint64_t)ldb * 2 < (1ull << 32);
    const bool mfma_ok = (K % GM_BK == 0) && (lda % 8 == 0) && (ldb % 8 == 0) && is_aligned(A, 16) && is_aligned(B, 16);
    if (impl == MAED_IMPL_VALU) return launch_valu<EPI, bf16>(A, lda, B, ldb, M, N, K, e, splitk, s);
    if (impl == MAED_IMPL_MFMA_GLDS1 || impl == MAED_IMPL_MFMA_GLDS2) {
        MAED_CHECK_ARG(mfma_ok && fits32, MAED_ERR_ALIGN, "gemm_nt(glds): need K%%64==0 (K=%lld), lda/ldb%%8==0, 16-B aligned A/B", (long long)K);
        return impl == MAED_IMPL_MFMA_GLDS1 ? launch_glds<EPI, 1>(A, lda, B, ldb, M, N, K, e, splitk, s)
                                            : launch_glds<EPI, 2>(A, lda, B, ldb, M, N, K, e, splitk, s);
    }
    const bool ok256 = mfma_ok && fits32 && K >= 128 && splitk == 1 && EPI != MAED_EPI_ATOMIC_F32;
    if (impl == MAED_IMPL_MFMA_256) {
        MAED_CHECK_ARG(ok256, MAED_ERR_ALIGN, "gemm_nt(256): need K%%64==0, K>=128 (K=%lld), lda/ldb%%8==0, 16-B aligned A/B, no split-K", (long long)K);
        maed_gemm_nt_256_launch(EPI, A, lda, B, ldb, M, N, K, e, s);
        return MAED_OK;
    }
    // persistent K-stream kernel (gemm_sk.hip): explicitly, or by the heuristic below
    const bool oksk = ok256 && maed_gemm_nt_sk_shape_ok(M, N, K) && !e.gn_sums;
    if (impl == MAED_IMPL_MFMA_SK) {
        const int mode = maed_opt(MAED_OPT_SK);
        MAED_CHECK_ARG(oksk && maed_gemm_nt_sk_launch(EPI, A, lda, B, ldb, M, N, K, e, mode == 0 ? 1 : mode, maed_opt(MAED_OPT_SK_GRID), s), MAED_ERR_ALIGN,
                       "gemm_nt(sk): need K%%128==0 (K=%lld), M, N >= 256, lda/ldb%%8==0, 16-B aligned A/B, no split-K, the library's slab allocation", (long long)K);
        return MAED_OK;
    }
    if (impl == MAED_IMPL_MFMA) {
        MAED_CHECK_ARG(mfma_ok, MAED_ERR_ALIGN, "gemm_nt(mfma): need K%%64==0 (K=%lld), lda/ldb%%8==0, 16-B aligned A/B", (long long)K);
        return launch_mfma<EPI>(A, lda, B, ldb, M, N, K, e, splitk, s);
    }
    if (!mfma_ok) return launch_valu<EPI, bf16>(A, lda, B, ldb, M, N, K, e, splitk, s);
    // measured at the cfg3 shapes (scripts/gemm_micro.py, profiles/r01_gemm_staging_variants_v3.txt): with the LDS-shuffled
    // epilogue the one-buffer kernel at 4 workgroups per CU wins or ties for every epilogue (fc1+GELU 110 vs 128 us, GELU'
    // 92 vs 112 us); the two-buffer kernel stays selectable (MAED_IMPL_MFMA_GLDS2); the split-K atomic epilogue keeps the
    // register-staged kernel
    // 256x256 pipelined tiles (gemm256.hip) where they measure faster (profiles/r02_gemm_tile_variants.txt): long K (fc2 85 -> 64 us,
    // d(qkv) 66 -> 47 us, 4096^3 964 -> 1250 TFLOP/s), or a grid that fits the chip in one round (proj 34 -> 32 us); with K = 512 and
    // many tiles (qkv, fc1) the four 128x128 workgroups per CU overlap their prologues / epilogues better than one 256x256 workgroup.
    // Never when the 256x256 grid would leave most CUs idle (stage-3 1x1 convolutions: 98 tiles).
    if constexpr (EPI != MAED_EPI_ATOMIC_F32) {
        const int64_t tiles256 = ((M + 255) / 256) * ((N + 255) / 256);
        // the persistent kernel wherever the 256x256 tiles can occupy most of the chip (stream-K spreads any tile count >= ~0.7 rounds over all CUs)
        const int skmode = maed_opt(MAED_OPT_SK);
        if (oksk && skmode != 0 && tiles256 * 10 >= (int64_t)maed_sk_cus() * 7 &&
            maed_gemm_nt_sk_launch(EPI, A, lda, B, ldb, M, N, K, e, skmode, maed_opt(MAED_OPT_SK_GRID), s))
            return MAED_OK;
        if (ok256 && N >= 256 && tiles256 >= 180 && (K >= 1024 || tiles256 <= 256)) {
            maed_gemm_nt_256_launch(EPI, A, lda, B, ldb, M, N, K, e, s);
            return MAED_OK;
        }
    }
    if constexpr (EPI == MAED_EPI_ATOMIC_F32) return launch_mfma<EPI>(A, lda, B, ldb, M, N, K, e, splitk, s);
    else return fits32 ? launch_glds<EPI, 1>(A, lda, B, ldb, M, N, K, e, splitk, s) : launch_mfma<EPI>(A, lda, B, ldb, M, N, K, e, splitk, s);
}

// 1x1 stride-1 convolution of the backbone on a channels_last activation viewed as (F*H*W, Cin) rows: y = x w^T (no bias: StdConv2dSame,
// resnetv2.py:74-93) = maed_gemm_nt's STORE epilogue, plus the GroupNorm statistics of the output for the GroupNorm that follows
extern "C" int maed_conv1x1_fwd(const void* x, int64_t ldx, const void* w, int64_t ldw, int64_t M, int Cout, int Cin, void* y, int64_t ldy, int hw,
                                double* gn_sums, int dtype, void* stream) {
    MAED_CHECK_ARG(x && w && y, MAED_ERR_ARG, "conv1x1_fwd: null pointer");
    const int np_call = maed_x3_take_dtype(dtype);
    MAED_CHECK_ARG(dtype == MAED_BF16 || dtype == MAED_F32, MAED_ERR_ARG, "conv1x1_fwd: bad dtype %d", dtype);
    MAED_CHECK_ARG(is_aligned(x, 16) && is_aligned(w, 16) && is_aligned(y, 16), MAED_ERR_ALIGN, "conv1x1_fwd: 16-B alignment");
    MAED_CHECK_ARG(!gn_sums || gn_stats_shape_ok(Cout, hw), MAED_ERR_SHAPE, "conv1x1_fwd: GroupNorm statistics need Cout = 32 * 2^k >= 64 and hw >= 128 (Cout=%d hw=%d)", Cout, hw);
    EpiArgs e{nullptr, y, ldy, nullptr, nullptr, 0, gn_sums, hw};
    if (dtype == MAED_F32) {        // fp32 operands on the split-bf16 MFMA kernel (gemm_x3.hip)
        const int x3np = np_call ? np_call : maed_x3_planes();
        MAED_CHECK_ARG(x3np, MAED_ERR_UNSUPPORTED, "conv1x1_fwd: f32 needs the split-bf16 matmul mode (maed_set_option(MAED_OPT_F32_MATMUL, 1 or 2))");
        MAED_CHECK_ARG(M >= 0 && Cout > 0 && Cin > 0 && ldx >= Cin && ldw >= Cin && ldy >= Cout && ldy % 4 == 0 && maed_x3_nt_shape_ok(x, ldx, w, ldw, Cin), MAED_ERR_SHAPE,
                       "conv1x1_fwd(f32): need Cin %% 32 == 0 and 4-element aligned strides (Cin=%d Cout=%d)", Cin, Cout);
        if (M == 0) return MAED_OK;
        MAED_PROPAGATE(maed_conv1x1_x3_launch(x3np, x, ldx, w, ldw, M, Cout, Cin, e, gn_sums != nullptr, (hipStream_t)stream));
        MAED_CHECK_LAUNCH("conv1x1_fwd(x3)");
        return MAED_OK;
    }
    MAED_CHECK_ARG(M >= 0 && Cout > 0 && Cin > 0 && Cin % GM_BK == 0 && ldx >= Cin && ldw >= Cin && ldy >= Cout && ldx % 8 == 0 && ldw % 8 == 0, MAED_ERR_SHAPE,
                   "conv1x1_fwd: need Cin %% 64 == 0 and 8-element aligned strides (Cin=%d Cout=%d)", Cin, Cout);
    MAED_CHECK_ARG((uint64_t)M * (uint64_t)ldx * 2 < (1ull << 32), MAED_ERR_SHAPE, "conv1x1_fwd: activation larger than 4 GB");
    if (M == 0) return MAED_OK;
    if (gn_sums) launch_glds<MAED_EPI_STORE, 1, true>(x, ldx, w, ldw, M, Cout, Cin, e, 1, (hipStream_t)stream);
    else launch_glds<MAED_EPI_STORE, 1>(x, ldx, w, ldw, M, Cout, Cin, e, 1, (hipStream_t)stream);
    MAED_CHECK_LAUNCH("conv1x1_fwd");
    return MAED_OK;
}

// maed_gemm_nt with the twin outputs of EpiArgs (fp32 operands on the split kernels, STORE / GELU epilogues): block.hip's twin forward
int maed_gemm_nt_twin(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K,
                      int dtype, int epilogue, const float* bias, void* out, int64_t ldo, void* out2,
                      const void* aux, int64_t ldaux, int splitk, int impl, void* stream, void* twin, bool out2_bf16, void* lo);
extern "C" int maed_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K,
                            int dtype, int epilogue, const float* bias, void* out, int64_t ldo, void* out2,
                            const void* aux, int64_t ldaux, int splitk, int impl, void* stream) {
    return maed_gemm_nt_twin(A, lda, B, ldb, M, N, K, dtype, epilogue, bias, out, ldo, out2, aux, ldaux, splitk, impl, stream, nullptr, false, nullptr);
}
int maed_gemm_nt_twin(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K,
                      int dtype, int epilogue, const float* bias, void* out, int64_t ldo, void* out2,
                      const void* aux, int64_t ldaux, int splitk, int impl, void* stream, void* twin, bool out2_bf16, void* lo) {
    // (twin + lo: the result as (hi, lo) bf16 planes for maed_gemm_nt_planes -- the fp32 form `out` may then be NULL)
    MAED_CHECK_ARG(A && B && (out || (twin && lo)), MAED_ERR_ARG, "gemm_nt: null pointer");
    MAED_CHECK_ARG(!lo || (twin && is_aligned(lo, 16)), MAED_ERR_ARG, "gemm_nt: a lo plane goes with the twin (hi) plane, 16-byte aligned");
    {   // MAED_F32X3 / MAED_F32X6: fp32 storage with an explicit engine = MAED_F32 + impl MAED_IMPL_X3 / _X6
        const int np_call = maed_x3_take_dtype(dtype);
        if (np_call && impl == MAED_IMPL_AUTO) impl = np_call == 2 ? MAED_IMPL_X3 : np_call == 1 ? MAED_IMPL_X1 : MAED_IMPL_X6;
    }
    MAED_CHECK_ARG(dtype == MAED_F32 || dtype == MAED_BF16, MAED_ERR_ARG, "gemm_nt: bad dtype %d", dtype);
    MAED_CHECK_ARG(M >= 0 && N > 0 && K > 0 && lda >= K && ldb >= K && ldo >= N, MAED_ERR_SHAPE, "gemm_nt: bad extents M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    if (splitk < 1) splitk = 1;
    MAED_CHECK_ARG(splitk == 1 || epilogue == MAED_EPI_ATOMIC_F32, MAED_ERR_ARG, "gemm_nt: split-K needs the atomic epilogue");
    if (M == 0) return MAED_OK;
    MAED_CHECK_ARG(!(twin || out2_bf16) || (dtype == MAED_F32 && (epilogue == MAED_EPI_STORE || epilogue == MAED_EPI_GELU) && N % 8 == 0 && ldo % 8 == 0
                                           && is_aligned(twin, 16) && (!out2_bf16 || is_aligned(out2, 16))),
                   MAED_ERR_ARG, "gemm_nt: twin outputs go with fp32 operands, the STORE / GELU epilogues, N and ldo multiples of 8 and 16-byte aligned buffers");
    EpiArgs e{bias, out, ldo, out2, aux, ldaux};
    e.twin = twin; e.out2_bf16 = out2_bf16; e.lo = lo;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    switch (epilogue) {
        case MAED_EPI_STORE: rc = dispatch<MAED_EPI_STORE>(A, lda, B, ldb, M, N, K, dtype, e, splitk, impl, s); break;
        case MAED_EPI_GELU:            // (out2 = NULL: the pre-activation is not stored -- inference)
            rc = dispatch<MAED_EPI_GELU>(A, lda, B, ldb, M, N, K, dtype, e, splitk, impl, s); break;
        case MAED_EPI_RESID_F32:
            MAED_CHECK_ARG(aux, MAED_ERR_ARG, "gemm_nt: RESID epilogue needs aux");
            rc = dispatch<MAED_EPI_RESID_F32>(A, lda, B, ldb, M, N, K, dtype, e, splitk, impl, s); break;
        case MAED_EPI_MUL_DGELU:
            MAED_CHECK_ARG(aux, MAED_ERR_ARG, "gemm_nt: MUL_DGELU epilogue needs aux");
            rc = dispatch<MAED_EPI_MUL_DGELU>(A, lda, B, ldb, M, N, K, dtype, e, splitk, impl, s); break;
        case MAED_EPI_ATOMIC_F32: rc = dispatch<MAED_EPI_ATOMIC_F32>(A, lda, B, ldb, M, N, K, dtype, e, splitk, impl, s); break;
        case MAED_EPI_STORE_F32: rc = dispatch<MAED_EPI_STORE_F32>(A, lda, B, ldb, M, N, K, dtype, e, splitk, impl, s); break;
        case MAED_EPI_TANH: rc = dispatch<MAED_EPI_TANH>(A, lda, B, ldb, M, N, K, dtype, e, splitk, impl, s); break;
        case MAED_EPI_ADD:
            MAED_CHECK_ARG(aux, MAED_ERR_ARG, "gemm_nt: ADD epilogue needs aux");
            rc = dispatch<MAED_EPI_ADD>(A, lda, B, ldb, M, N, K, dtype, e, splitk, impl, s); break;
        default: maed_set_error("gemm_nt: bad epilogue %d", epilogue); return MAED_ERR_ARG;
    }
    if (rc != MAED_OK) return rc;
    MAED_CHECK_LAUNCH("gemm_nt");
    return MAED_OK;
}
