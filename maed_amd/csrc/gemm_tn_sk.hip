// Persistent K-stream WEIGHT-GRADIENT GEMM (round 6):  dW[N,K] += Y[M,N]^T . X[M,K]  (+ dbias[N] += colsum(Y)), bf16 operands, fp32 accumulation -- the
// reduction runs over the ROWS of both operands (nn.Linear / 1x1-convolution backward through autograd: vision_transformer.py:98-111,124-128,147,176;
// resnetv2.py:91-93).  The "roofline" kernel of bench.py: rounds 3-5 left it at 0.19 of the MFMA peak on 128 x 128 tiles (gemm_tn.hip, gemm_tn2.hip), bound by the
// L2 -> LDS traffic of that tile and by 16-64 MB of closing fp32 atomics per launch; a 256 x 256 tile on the same structure lost to its own atomics.
//
// Here the long-K habitat of stream-K: few output tiles (qkv: 6 x 2 of 256 x 256), a reduction of 25 216 rows = 197 pairs of 64-row K tiles.
//   * The workgroups of the launch (one per CU) are dealt to the output tiles -- tile t gets the workgroups [ceil(t G / T), ceil((t+1) G / T)) -- and the
//     workgroups of a tile share its rows evenly at K-tile-PAIR granularity (21 or 22 workgroups x 9 or 10 pairs at qkv).
//   * A workgroup runs ONE item through gemm_sk.hip's pipeline: the same ring of eight 16-KB half-tile slots, four phases per K tile, LDS-DMA issue order =
//     consumption order six to seven phases ahead, counted vmcnt, waves 4-7 one barrier behind waves 0-3.  What differs is the operand path: both operands are copied
//     UNCHANGED (row-major [64 reduction rows][128 columns] half-tiles, 16-byte chunk c of row m at slot c ^ ((m & 3) << 2), swizzled on the copy's source address)
//     and the MFMA fragments -- eight reduction rows of one column per lane -- come out of that image by ds_read_b64_tr_b16 (gemm_tn2.hip's layout: two reads per
//     fragment, conflict-free).
//   * No atomics: every workgroup stores its fp32 partial tile to a slab (256 KB, tile-linear, full 256-byte row segments through the LDS staging of gemm_sk.hip)
//     and a second launch (gemm_tn_sk_reduce_kernel) adds the slabs of each tile in a FIXED order onto dW -- deterministic, 64 MB of plain coalesced traffic instead
//     of 16-64 MB of contended atomics, and no inter-workgroup waiting inside a launch (two such kernels may share the chip).
//   * Bias gradient: the workgroups of the first column of tiles sum the Y fragments they hold anyway (v_dot2c_f32_bf16 against ones in the waves wc == 0, in the
//     shadow of the MFMAs) into 256 extra floats behind their slab; the reduce launch folds them.
// Output tile coordinates: n = n0 + qm * 128 + wr * 64 + rt * 32 + (lane & 31)  (both operands' half-tiles are CONTIGUOUS 128-column windows of Y / X, so a wave's
// two row / column blocks sit 128 apart), k = k0 + qn * 128 + wc * 32 + i.  First MFMA operand = the X fragment: a lane owns one output ROW (n) and 4 consecutive k.
#include "common.cuh"
#include "gemm_epilogue.cuh"
#include <mutex>

#define TS_T 256
#define TS_BK 64
#define TS_SLOT (TS_BK * 128)          // elements per half-tile slot (16 KB): [64 reduction rows][128 columns]
#define TS_A0 0
#define TS_B0 1
#define TS_B1 2
#define TS_A1 3
#define TS_RING_ELEMS (2 * 4 * TS_SLOT)
#define TS_LDS_ELEMS (TS_RING_ELEMS + 4 * 2048)
#define TS_STAGE_BYTE0 (7 * TS_SLOT * 2)
#define TS_SLAB_FLOATS (TS_T * TS_T + TS_T)        // partial tile + 256 partial column sums of Y
#define TS_UNI(x_) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x_)))
#ifdef MAED_HOSTSIM
#define TS_OPAQUE(v_) ((void)0)
#else
#define TS_OPAQUE(v_) asm volatile("" : "+v"(v_))
#endif

// sum of the eight bf16 of a fragment, added to acc: four v_dot2c_f32_bf16 against (1, 1).  Inline assembly, volatile: as ordinary code hipcc gathers the pair's 128
// dot products behind its last phase and keeps all 32 fragments alive for them (364 bytes of scratch inside the K stream); the accumulating form (dst = src2, same
// opcode back to back) needs no wait states, a DIFFERENT VALU instruction reading the sum would need three -- the closing s_nop 2 (gfx940 DOT hazards; the
// compiler's hazard recognizer does not look inside an asm statement).
__device__ __forceinline__ float ts_sum8(const bf16x8_t& f, float acc) {
    uint4 u;
    __builtin_memcpy(&u, &f, 16);
#ifdef MAED_HOSTSIM
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
    for (int i = 0; i < 4; ++i) acc += __uint_as_float(w[i] << 16) + __uint_as_float(w[i] & 0xffff0000u);
    return acc;
#else
    asm volatile("v_dot2c_f32_bf16 %0, 0x3f803f80, %1\n\tv_dot2c_f32_bf16 %0, 0x3f803f80, %2\n\tv_dot2c_f32_bf16 %0, 0x3f803f80, %3\n\t"
                 "v_dot2c_f32_bf16 %0, 0x3f803f80, %4\n\ts_nop 2"
                 : "+v"(acc) : "v"(u.x), "v"(u.y), "v"(u.z), "v"(u.w));
    return acc;
#endif
}

struct TsPlan {
    int tiles, tiles_k;        // output tiles, tiles per row of tiles (K direction)
    int pairs;                 // K-tile pairs of the whole reduction (M / 128)
};
// the workgroups of tile t: [ts_first(t), ts_first(t + 1))
__host__ __device__ __forceinline__ int ts_first(int t, int tiles, int grid) { return (int)(((int64_t)t * grid + tiles - 1) / tiles); }

template <bool BIAS>
__global__ __launch_bounds__(512, 2) void gemm_tn_sk_bf16_kernel(const bf16* __restrict__ Y, int64_t ldy, const bf16* __restrict__ X, int64_t ldx,
                                                                  int N, int K, TsPlan P, float* __restrict__ slabs) {
    constexpr int want_bias = BIAS ? 1 : 0;
    __shared__ __attribute__((aligned(1024))) unsigned short lds_raw[TS_LDS_ELEMS];          // 144 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5, i16 = lane & 15;
    const int grid = (int)gridDim.x, g = (int)blockIdx.x;
    // ---- this workgroup's item: tile t, pairs [p0, p0 + np) of its reduction
    const int t = (int)TS_UNI((int)(((int64_t)g * P.tiles) / grid));
    const int g0 = ts_first(t, P.tiles, grid), nw = ts_first(t + 1, P.tiles, grid) - g0, idx = g - g0;
    const int p0 = (int)TS_UNI((int)(((int64_t)idx * P.pairs) / nw)), np = (int)TS_UNI((int)(((int64_t)(idx + 1) * P.pairs) / nw)) - p0;
    const int n0 = (t / P.tiles_k) * TS_T, k0 = (t % P.tiles_k) * TS_T;
    float* const slab = slabs + (int64_t)g * TS_SLAB_FLOATS;
    if (np <= 0) {            // more workgroups than pairs (never at the shapes the launcher sends): an all-zero slab keeps the reduce launch uniform
        for (int i = tid; i < TS_SLAB_FLOATS; i += 512) slab[i] = 0.f;
        return;
    }

    // ---- staging map: a half-tile is 64 reduction rows x 256 B = 1024 chunks of 16 B, two per thread (round i = 0, 1); a copy instruction covers four rows:
    //      wave w, round i -> rows 4 (w + 8 i) .. +3, lane l -> row + (l >> 4), chunk SLOT l & 15; the chunk that belongs there undoes the swizzle.  Columns past
    //      N / K (multiples of 8, not of the tile) copy the tile's first chunk instead: every lane takes part in every copy, the duplicate lands in columns whose
    //      products are never reduced.
    uint32_t ao0, ao1, ao2, ao3, bo0, bo1, bo2, bo3;                  // index 2 * i + q: BYTE offsets inside a K tile (launcher: < 2^31)
    {
        const uint32_t ldy2 = (uint32_t)ldy * 2u, ldx2 = (uint32_t)ldx * 2u;
#define TS_OFFS(j)                                                                                  \
        {                                                                                           \
            const int i_ = (j) >> 1, q_ = (j) & 1;                                                  \
            const int row = 4 * (wave + 8 * i_) + (lane >> 4), chunk = (lane & 15) ^ ((row & 3) << 2); \
            int ca = n0 + q_ * 128 + chunk * 8, cb = k0 + q_ * 128 + chunk * 8;                     \
            if (ca >= N) ca = n0;                                                                   \
            if (cb >= K) cb = k0;                                                                   \
            ao##j = (uint32_t)row * ldy2 + (uint32_t)ca * 2u; bo##j = (uint32_t)row * ldx2 + (uint32_t)cb * 2u; \
        }
        TS_OFFS(0) TS_OFFS(1) TS_OFFS(2) TS_OFFS(3)
#undef TS_OFFS
    }
    unsigned short* const ldsw = lds_raw + wave * 512;                // this wave's four rows of round 0 inside a slot (scalar); round 1: + 8 * 512
    const int64_t ktA = (int64_t)TS_BK * ldy * 2, ktB = (int64_t)TS_BK * ldx * 2;      // bytes per K tile
    const char* const Yb = reinterpret_cast<const char*>(Y);
    const char* const Xb = reinterpret_cast<const char*>(X);
    // one LDS-DMA per thread: scalar base (operand + K-tile offset) + 32-bit lane offset -> round i_ of slot slot_ of buffer buf_
#define TS_DMA(base_, kt_, step_, off_, buf_, slot_, i_) MAED_LDS_DMA16((base_) + (int64_t)TS_UNI(kt_) * (step_), off_, ldsw + ((buf_) * 4 + (slot_)) * TS_SLOT + (i_) * 8 * 512)

    // ---- fragments by transposing reads (gemm_tn2.hip): inside a 16-lane group lane 4 j + t reads row j, columns 4 t .. 4 t + 3 and receives column 4 (lane >> 2)...
    //      i.e. lane l31 ends up with column l31 of a 32-column block, rows 4 hi + (0..3) of a 16-row step from the first read and + 8 from the second
    const int prow = 4 * hi + (i16 >> 2), pcol = (lane & 16) + 4 * (i16 & 3), psw = (i16 >> 2) << 2;
    const char* const ldsb = reinterpret_cast<const char*>(lds_raw);
    auto frag_base = [&](int c) { return ldsb + prow * 256 + ((((c + pcol) >> 3) ^ psw) << 4) + ((pcol & 7) << 1); };
    const char* const fa0 = frag_base(wr * 64);            // Y block rt = 0 of this wave's 64 columns inside an A half-tile
    const char* const fa1 = frag_base(wr * 64 + 32);
    const char* const fb = frag_base(wc * 32);             // X block of this wave inside a B half-tile
    union Frag { bf16x8_t v; uint2 u[2]; };
    bf16x8_t a00, a01, a02, a03, a10, a11, a12, a13;        // a[rt][kk]
    bf16x8_t b00, b01, b02, b03, b10, b11, b12, b13;        // b[qn][kk]
    f32x16_t c000, c001, c010, c011, c100, c101, c110, c111;    // c[qm][rt][qn]
#pragma unroll
    for (int x = 0; x < 16; ++x) { c000[x] = 0.f; c001[x] = 0.f; c010[x] = 0.f; c011[x] = 0.f; c100[x] = 0.f; c101[x] = 0.f; c110[x] = 0.f; c111[x] = 0.f; }
    float bs00 = 0.f, bs01 = 0.f, bs10 = 0.f, bs11 = 0.f;    // partial column sums of Y: bs[qm][rt], this lane's column, its 8 of every 16 rows
    const bool do_bias = want_bias && (t % P.tiles_k) == 0 && wc == 0;          // wave-uniform
#define TS_TR(dst_, base_, buf_, slot_, kk_)                                                                                    \
    {                                                                                                                           \
        const auto lo__ = MAED_DS_READ_TR16((base_) + (((buf_) * 4 + (slot_)) * TS_SLOT) * 2 + (kk_) * 16 * 256);               \
        const auto hi__ = MAED_DS_READ_TR16((base_) + (((buf_) * 4 + (slot_)) * TS_SLOT) * 2 + ((kk_) * 16 + 8) * 256);         \
        Frag f__;                                                                                                               \
        __builtin_memcpy(&f__.u[0], &lo__, 8); __builtin_memcpy(&f__.u[1], &hi__, 8);                                           \
        dst_ = f__.v;                                                                                                           \
    }
#define TS_READ_A(buf_, slot_)                                                                                                  \
    TS_TR(a00, fa0, buf_, slot_, 0) TS_TR(a01, fa0, buf_, slot_, 1) TS_TR(a02, fa0, buf_, slot_, 2) TS_TR(a03, fa0, buf_, slot_, 3) \
    TS_TR(a10, fa1, buf_, slot_, 0) TS_TR(a11, fa1, buf_, slot_, 1) TS_TR(a12, fa1, buf_, slot_, 2) TS_TR(a13, fa1, buf_, slot_, 3)
#define TS_READ_B0(buf_) TS_TR(b00, fb, buf_, TS_B0, 0) TS_TR(b01, fb, buf_, TS_B0, 1) TS_TR(b02, fb, buf_, TS_B0, 2) TS_TR(b03, fb, buf_, TS_B0, 3)
#define TS_READ_B1(buf_) TS_TR(b10, fb, buf_, TS_B1, 0) TS_TR(b11, fb, buf_, TS_B1, 1) TS_TR(b12, fb, buf_, TS_B1, 2) TS_TR(b13, fb, buf_, TS_B1, 3)
    // column sums of the A fragments just read (once per A half-tile: the phases that read A), in the shadow of the phase's MFMAs
#define TS_BIAS(s0_, s1_)                                                                                                       \
    if constexpr (BIAS) {      /* (every wave, no branch inside the K stream: 32 v_dot2c per A half-tile beside 16 MFMAs; only the owners store them) */ \
        s0_ = ts_sum8(a00, s0_); s0_ = ts_sum8(a01, s0_); s0_ = ts_sum8(a02, s0_); s0_ = ts_sum8(a03, s0_);                     \
        s1_ = ts_sum8(a10, s1_); s1_ = ts_sum8(a11, s1_); s1_ = ts_sum8(a12, s1_); s1_ = ts_sum8(a13, s1_);                     \
    }
    // one phase (gemm_sk.hip / gemm256.hip): fragment reads ; barrier ; landed ; 8 MFMAs at raised priority with two LDS-DMAs in their shadow ; counted wait ; barrier
#define TS_PHASE(READS_, ISSUE0_, ISSUE1_, WAIT_, c0_, c1_, bq_, EXTRA_)                             \
    READS_                                                                                          \
    __builtin_amdgcn_s_barrier();                                                                   \
    MAED_WAIT_LGKMCNT0();                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    __builtin_amdgcn_s_setprio(1);                                                                  \
    c0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##0, a00, c0_, 0, 0, 0);                   \
    c1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##0, a10, c1_, 0, 0, 0);                   \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    ISSUE0_;                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    c0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##1, a01, c0_, 0, 0, 0);                   \
    c1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##1, a11, c1_, 0, 0, 0);                   \
    c0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##2, a02, c0_, 0, 0, 0);                   \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    ISSUE1_;                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    c1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##2, a12, c1_, 0, 0, 0);                   \
    c0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##3, a03, c0_, 0, 0, 0);                   \
    c1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##3, a13, c1_, 0, 0, 0);                   \
    EXTRA_                                                                                          \
    __builtin_amdgcn_sched_barrier(0);     /* (the column sums stay in the phase whose fragments they read: sunk to the end of the pair they keep 128 registers alive) */ \
    __builtin_amdgcn_s_setprio(0);                                                                  \
    WAIT_;                                                                                          \
    __builtin_amdgcn_s_barrier();
#define TS_NONE ((void)0)
    // K-tile pair (t, t+1) of the stream, t in buffer 0 (gemm_sk.hip): even tile q1: A1(t+1)   q2..q4: A0, B0, B1 of t+2;  odd tile q1: A1(t+2)   q2..q4: A0, B0, B1 of t+3.
    // kt_nx: K tile index of the pair to issue next (stays on the last pair at the end of the stream: its copies are repeated into slots nobody reads again).
#define TS_PAIR(EVENWAIT_)                                                                                                                                \
    {                                                                                                                                                     \
        const uint32_t k1__ = kt_nx + 1;                                                                                                                  \
        TS_PHASE(TS_READ_A(0, TS_A0) TS_READ_B0(0), TS_DMA(Yb, kpa, ktA, ao1, 1, TS_A1, 0), TS_DMA(Yb, kpa, ktA, ao3, 1, TS_A1, 1), EVENWAIT_, c000, c010, b0, TS_BIAS(bs00, bs01)) \
        TS_PHASE(TS_READ_B1(0), TS_DMA(Yb, kt_nx, ktA, ao0, 0, TS_A0, 0), TS_DMA(Yb, kt_nx, ktA, ao2, 0, TS_A0, 1), EVENWAIT_, c001, c011, b1, )          \
        TS_PHASE(TS_READ_A(0, TS_A1), TS_DMA(Xb, kt_nx, ktB, bo0, 0, TS_B0, 0), TS_DMA(Xb, kt_nx, ktB, bo2, 0, TS_B0, 1), EVENWAIT_, c101, c111, b1, TS_BIAS(bs10, bs11)) \
        TS_PHASE(, TS_DMA(Xb, kt_nx, ktB, bo1, 0, TS_B1, 0), TS_DMA(Xb, kt_nx, ktB, bo3, 0, TS_B1, 1), EVENWAIT_, c100, c110, b0, )                        \
        TS_PHASE(TS_READ_A(1, TS_A0) TS_READ_B0(1), TS_DMA(Yb, kt_nx, ktA, ao1, 0, TS_A1, 0), TS_DMA(Yb, kt_nx, ktA, ao3, 0, TS_A1, 1), MAED_WAIT_VMCNT(8), c000, c010, b0, TS_BIAS(bs00, bs01)) \
        TS_PHASE(TS_READ_B1(1), TS_DMA(Yb, k1__, ktA, ao0, 1, TS_A0, 0), TS_DMA(Yb, k1__, ktA, ao2, 1, TS_A0, 1), MAED_WAIT_VMCNT(8), c001, c011, b1, )    \
        TS_PHASE(TS_READ_A(1, TS_A1), TS_DMA(Xb, k1__, ktB, bo0, 1, TS_B0, 0), TS_DMA(Xb, k1__, ktB, bo2, 1, TS_B0, 1), MAED_WAIT_VMCNT(8), c101, c111, b1, TS_BIAS(bs10, bs11)) \
        TS_PHASE(, TS_DMA(Xb, k1__, ktB, bo1, 1, TS_B1, 0), TS_DMA(Xb, k1__, ktB, bo3, 1, TS_B1, 1), MAED_WAIT_VMCNT(8), c100, c110, b0, )                 \
        kpa = k1__;                                                                                                                                       \
    }
    // (the A1 offsets are the same for every K tile of the item: ao1 / ao3 serve the pending A1 as well)

    // ---- prologue: the first pair except the A1 half of its odd tile; everything landed before the first (wait-free) pair
    uint32_t kt_nx = (uint32_t)(2 * p0), kpa;
    int left = np;
    {
        const uint32_t k1 = kt_nx + 1;
        TS_DMA(Yb, kt_nx, ktA, ao0, 0, TS_A0, 0); TS_DMA(Yb, kt_nx, ktA, ao2, 0, TS_A0, 1);
        TS_DMA(Xb, kt_nx, ktB, bo0, 0, TS_B0, 0); TS_DMA(Xb, kt_nx, ktB, bo2, 0, TS_B0, 1);
        TS_DMA(Xb, kt_nx, ktB, bo1, 0, TS_B1, 0); TS_DMA(Xb, kt_nx, ktB, bo3, 0, TS_B1, 1);
        TS_DMA(Yb, kt_nx, ktA, ao1, 0, TS_A1, 0); TS_DMA(Yb, kt_nx, ktA, ao3, 0, TS_A1, 1);
        TS_DMA(Yb, k1, ktA, ao0, 1, TS_A0, 0); TS_DMA(Yb, k1, ktA, ao2, 1, TS_A0, 1);
        TS_DMA(Xb, k1, ktB, bo0, 1, TS_B0, 0); TS_DMA(Xb, k1, ktB, bo2, 1, TS_B0, 1);
        TS_DMA(Xb, k1, ktB, bo1, 1, TS_B1, 0); TS_DMA(Xb, k1, ktB, bo3, 1, TS_B1, 1);
        kpa = k1;
    }
#define TS_ADVANCE() { if (left > 1) { --left; kt_nx = TS_UNI(kt_nx + 2); } }
    TS_ADVANCE()
    MAED_WAIT_VMCNT0();
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();        // waves 4-7 run one barrier behind

    TS_PAIR(TS_NONE)
    TS_ADVANCE()
    for (int p = 1; p < np; ++p) {
        TS_PAIR(MAED_WAIT_VMCNT(8))
        TS_ADVANCE()
    }
    __builtin_amdgcn_sched_barrier(0);
    MAED_WAIT_VMCNT0();        // the repeated copies of the last pair have landed (none targets the staging slot, but the LDS goes to the next workgroup at the end)

    // ---- the partial tile -> slab, tile-linear [256 n][256 k] fp32, through the wave-private staging of gemm_sk.hip (16 rows x 256 B, chunk c of row r at slot c ^ r;
    //      waves 0-3 in slot A1 of buffer 1, waves 4-7 behind the ring): a piece is 16 rows x (32 + 32) columns -- the wave's two column blocks sit 128 apart
    char* const stg = reinterpret_cast<char*>(lds_raw) + TS_STAGE_BYTE0 + wave * 4096;
    int r16 = l31 & 15, rr = lane >> 3, c8 = lane & 7;
    const int rhalf = l31 >> 4;
    TS_OPAQUE(r16); TS_OPAQUE(rr); TS_OPAQUE(c8);
    const int kcol = (c8 < 4 ? wc * 32 + c8 * 8 : 128 + wc * 32 + (c8 - 4) * 8);
#define TS_STORE_HALF(accA_, accB_, qm_, rt_, h_)                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                                      \
    MAED_WAVE_LDS_SYNC();                                                                                                   \
    if (rhalf == (h_)) {                                                                                                    \
        _Pragma("unroll") for (int q4 = 0; q4 < 4; ++q4) {                                                                  \
            *reinterpret_cast<float4*>(stg + r16 * 256 + (((2 * q4 + hi) ^ r16) << 4)) = make_float4(accA_[4 * q4], accA_[4 * q4 + 1], accA_[4 * q4 + 2], accA_[4 * q4 + 3]);      \
            *reinterpret_cast<float4*>(stg + r16 * 256 + (((8 + 2 * q4 + hi) ^ r16) << 4)) = make_float4(accB_[4 * q4], accB_[4 * q4 + 1], accB_[4 * q4 + 2], accB_[4 * q4 + 3]);  \
        }                                                                                                                   \
    }                                                                                                                       \
    MAED_WAVE_LDS_SYNC();                                                                                                   \
    _Pragma("unroll") for (int ps = 0; ps < 2; ++ps) {                                                                      \
        const int lr = ps * 8 + rr;                                                                                         \
        const int nrow = (qm_) * 128 + wr * 64 + (rt_) * 32 + (h_) * 16 + lr;                                               \
        const float4 u0 = *reinterpret_cast<const float4*>(stg + lr * 256 + (((2 * c8) ^ lr) << 4));                        \
        const float4 u1 = *reinterpret_cast<const float4*>(stg + lr * 256 + (((2 * c8 + 1) ^ lr) << 4));                    \
        float* const dst = slab + nrow * TS_T + kcol;                                                                       \
        *reinterpret_cast<float4*>(dst) = u0; *reinterpret_cast<float4*>(dst + 4) = u1;                                     \
    }
#define TS_STORE_PIECE(accA_, accB_, qm_, rt_) TS_STORE_HALF(accA_, accB_, qm_, rt_, 0) TS_STORE_HALF(accA_, accB_, qm_, rt_, 1)
    TS_STORE_PIECE(c000, c001, 0, 0)
    TS_STORE_PIECE(c010, c011, 0, 1)
    TS_STORE_PIECE(c100, c101, 1, 0)
    TS_STORE_PIECE(c110, c111, 1, 1)
    // ---- the column sums of Y: fold the two lane halves (rows 4 hi + ... of every 16), then one lane per column stores
    if (want_bias && wc == 0) {
        float v00 = bs00, v01 = bs01, v10 = bs10, v11 = bs11;
        if (!do_bias) { v00 = 0.f; v01 = 0.f; v10 = 0.f; v11 = 0.f; }
        v00 += __shfl_xor(v00, 32); v01 += __shfl_xor(v01, 32); v10 += __shfl_xor(v10, 32); v11 += __shfl_xor(v11, 32);
        if (hi == 0) {
            float* const bp = slab + TS_T * TS_T + wr * 64 + l31;
            bp[0] = v00; bp[32] = v01; bp[128] = v10; bp[160] = v11;
        }
    }
}

// dW[n][k] += sum over the workgroups of the tile (ascending: a fixed order) of their slabs; dbias[n] += the same over the first column of tiles.
// One thread: 4 consecutive k of one row; a workgroup of 256: 4 rows x 256 k.
__global__ __launch_bounds__(256) void gemm_tn_sk_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dW, int64_t ldw, float* __restrict__ dbias,
                                                                int N, int K, TsPlan P, int grid) {
    const int t = (int)blockIdx.y, rb = (int)blockIdx.x;                  // tile, block of 4 rows
    const int g0 = ts_first(t, P.tiles, grid), g1 = ts_first(t + 1, P.tiles, grid);
    const int n0 = (t / P.tiles_k) * TS_T, k0 = (t % P.tiles_k) * TS_T;
    const int r = rb * 4 + ((int)threadIdx.x >> 6), c = ((int)threadIdx.x & 63) * 4;
    const int n = n0 + r, k = k0 + c;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    // (eight independent loads in flight per thread: a slab hop is a round trip to the Infinity Cache / HBM, ~20 of them in a row would be latency, not bandwidth)
#pragma unroll 8
    for (int g = g0; g < g1; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(slabs + (int64_t)g * TS_SLAB_FLOATS + r * TS_T + c);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (n < N && k < K) {                 // (K % 4 == 0: launcher)
        float4* const d = reinterpret_cast<float4*>(dW + (int64_t)n * ldw + k);
        float4 o = *d;
        o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
        *d = o;
    }
    if (dbias && (t % P.tiles_k) == 0 && rb == 0 && threadIdx.x < TS_T) {          // the tile's 256 column sums: one thread each (blocks rb == 0 only)
        const int nn = n0 + (int)threadIdx.x;
        float b = 0.f;
#pragma unroll 8
        for (int g = g0; g < g1; ++g) b += slabs[(int64_t)g * TS_SLAB_FLOATS + TS_T * TS_T + threadIdx.x];
        if (nn < N) dbias[nn] += b;
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------------------
// Slabs: the library's one device allocation (gemm_sk.hip maed_sk_init: SK_SLOTS sets of one slab per CU, a set per stream that launches these kernels).
float* maed_sk_slab_set(hipStream_t s, size_t* bytes, int* ncu);       // gemm_sk.hip

bool maed_gemm_tn_sk_ok(int64_t M, int N, int K, int64_t ldy, int64_t ldx, int64_t ldw, const void* Y, const void* X, const void* dW) {
    // a tile is worth its 256 x 256 accumulators only when most of it is output: N, K at least 256 (the STE's linears, stage-3 convolutions); ragged edges are fine
    return M >= 128 && M % 128 == 0 && N >= 256 && K >= 256 && N % 8 == 0 && K % 8 == 0 && ldy % 8 == 0 && ldx % 8 == 0 && ldw % 4 == 0
           && (M * (ldy > ldx ? ldy : ldx) * 2) < (1ll << 40) && (int64_t)TS_BK * (ldy > ldx ? ldy : ldx) * 2 + 65536 < (1ll << 31)
           && is_aligned(Y, 16) && is_aligned(X, 16) && is_aligned(dW, 16);
}
int maed_gemm_tn_sk_launch(const void* Y, int64_t ldy, const void* X, int64_t ldx, int64_t M, int N, int K, float* dW, int64_t ldw, float* dbias, int grid_opt,
                           hipStream_t s) {
    size_t bytes = 0;
    int ncu = 0;
    float* slabs = maed_sk_slab_set(s, &bytes, &ncu);
    if (!slabs) return MAED_ERR_UNSUPPORTED;
    TsPlan P;
    const int tn = (N + TS_T - 1) / TS_T, tk = (K + TS_T - 1) / TS_T;
    P.tiles = tn * tk; P.tiles_k = tk; P.pairs = (int)(M / 128);
    int G = grid_opt > 0 && grid_opt < ncu ? grid_opt : ncu;
    if ((size_t)G * TS_SLAB_FLOATS * sizeof(float) > bytes) G = (int)(bytes / (TS_SLAB_FLOATS * sizeof(float)));
    if (P.tiles > G || P.pairs < 1) return MAED_ERR_UNSUPPORTED;
    // Measured (profiles/r06_tn_sk_micro.txt): the slabs cost a fixed ~25-30 us per launch (64 MB written at the end of the stream by every workgroup at once, 64 MB
    // read back by the reduce launch) -- it pays from ~32 K-tile pairs per workgroup (cfg5's fc1 / fc2 weight gradients: 187 vs 233 us), not at cfg3's 9-12 pairs
    // (qkv 76 vs 66 us).  grid_opt > 0 (tests, sweeps) takes the kernel regardless.
    if (grid_opt <= 0 && (int64_t)P.tiles * P.pairs < (int64_t)32 * G) return MAED_ERR_UNSUPPORTED;
    if (dbias) hipLaunchKernelGGL(gemm_tn_sk_bf16_kernel<true>, dim3((unsigned)G), dim3(512), 0, s, (const bf16*)Y, ldy, (const bf16*)X, ldx, N, K, P, slabs);
    else hipLaunchKernelGGL(gemm_tn_sk_bf16_kernel<false>, dim3((unsigned)G), dim3(512), 0, s, (const bf16*)Y, ldy, (const bf16*)X, ldx, N, K, P, slabs);
    hipLaunchKernelGGL(gemm_tn_sk_reduce_kernel, dim3(TS_T / 4, (unsigned)P.tiles), dim3(256), 0, s, (const float*)slabs, dW, ldw, dbias, N, K, P, G);
    return MAED_OK;
}
