// nn.Linear family on 256x256x64 tiles with a counted-vmcnt LDS-DMA pipeline (bf16, fp32 accumulation): the kernel behind
// maed_gemm_nt for the large-M shapes of the STE (vision_transformer.py:98-111,124-128,147,176) and the wide 1x1 convolutions.
//
// Why a second tile size.  Measured ablation of the 128x128 kernel at the qkv shape (profiles/r02_gemm_ablation.txt): loads alone
// 28 us (21 TB/s L2->LDS: the load path is saturated), MFMA + fragment reads alone 28 us, stores alone 18 us -- and 59 us together:
// the three barely overlap, because with 32 KB of LDS per workgroup only ~64 KB of loads are in flight per CU, and a 128x128 tile
// needs 64 B/clk/CU of operand traffic at the MFMA rate.  A 256x256 tile needs half of that, and the schedule below keeps 64 KB of
// LDS-DMA in flight per CU at every moment of the main loop.
//
// Geometry.  8 waves = 2 (M) x 4 (N); a wave owns 128 x 64 of the output = 4 quadrants of 64 x 32 (2 MFMA 32x32x16 tiles each),
// 128 accumulator registers.  A K tile (BK = 64) is four 16-KB HALF-TILES: A0/A1 = the rows every wave needs for its quadrant row
// qm = 0/1 (128 rows), B0/B1 = the weight rows for quadrant column qn = 0/1 (128 rows).  LDS = 2 buffers x 4 half-tile slots = 128 KB,
// each slot a [128][64] bf16 image, 16-B chunk index XOR-swizzled with (row>>1)&7 on the LDS-DMA SOURCE address (the DMA writes
// base + lane*16) and on the ds_read_b128 fragment reads: conflict-free.
//
// Schedule (one "phase" = one quadrant: fragment reads, barrier, 8 MFMAs with one half-tile of LDS-DMA issued in their shadow, counted
// wait, barrier).  K tile t, phases q = 1..4:
//     q   reads (ds_read_b128)   MFMA quadrant   LDS-DMA issued between the MFMAs (2 per thread)
//     1   A0 (8) + B0 (4)        (0,0)           A1 of tile t+1
//     2   B1 (4)                 (0,1)           A0 of tile t+2
//     3   A1 (8)                 (1,1)           B0 of tile t+2
//     4   --                     (1,0)           B1 of tile t+2
// Every phase ends with s_waitcnt vmcnt(8): the four most recent half-tiles (64 KB) stay in flight, the one issued four phases ago has
// landed; it is read two phases later at the earliest, and a slot is re-targeted by a DMA one phase after its last read at the earliest
// -- what LDS-DMA needs under staggered wave groups (MI355X_MICROARCH.md: nothing orders a ds_read behind a pending LDS-DMA except the
// issuing wave's counted vmcnt plus a barrier).  Waves 4-7 run one barrier behind waves 0-3 (one wave of each group per SIMD): while
// one group issues its MFMAs the other reads fragments, and s_setprio favours the group in its MFMA segment.  The tail drains with
// vmcnt(6/4/2/0).  Raw s_barrier only (a __syncthreads() would drain the DMA queue).
//
// Epilogue: the fused epilogues of gemm_epilogue.cuh through the same LDS shuffle as the 128x128 kernel (a lane owns one output row in
// the transposed accumulators; each wave parks 32 x 64 fp32 in LDS and stores full 128/256-byte row segments).
#include "common.cuh"
#include "gemm_epilogue.cuh"

#define G2_T 256
#define G2_BK 64
#define G2_SLOT (128 * G2_BK)      // elements per half-tile slot (16 KB)
// slot order inside a buffer: consumption order
#define G2_A0 0
#define G2_B0 1
#define G2_B1 2
#define G2_A1 3

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_256_bf16_kernel(const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ B, int64_t ldb,
                                                                   int64_t M, int64_t N, int64_t K, int tiles_n, EpiArgs e
#ifdef MAED_GEMM_ABLATE
                                                                   , int ablate    // diagnostic build only: 1 no stores, 2 no loads, 4 no MFMA / fragment reads
#endif
                                                                   ) {
#ifndef MAED_GEMM_ABLATE
    constexpr int ablate = 0;
#endif
    __shared__ __attribute__((aligned(1024))) unsigned short lds_raw[2 * 4 * G2_SLOT];          // 128 KB; the epilogue re-uses it
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // scalar: LDS-DMA bases (M0) and the wave-group branches stay on the SALU
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(id / tiles_n) * G2_T, n0 = (int64_t)(id % tiles_n) * G2_T;
    const int nkt = (int)(K / G2_BK);                                                              // >= 2 (host-checked)

    // ---- staging map: a half-tile is 128 rows x 128 B = 1024 chunks of 16 B, two per thread (round i = 0, 1); wave w fills slot rows
    //      8w + 64i .. +7, lane l the (swizzled) chunk of row 8w + 64i + (l>>3).  Slot row s of A-half q is tile row (s>>6)*128 + q*64 + (s&63)
    //      (the 64 rows of quadrant row q of M-wave s>>6); slot row s of B-half q is tile column (s>>5)*64 + q*32 + (s&31).
    const int r = wave * 8 + (lane >> 3);                                                          // slot row of round 0
    const int schunk = (lane & 7) ^ ((r >> 1) & 7);
    uint32_t ao0, ao1, ao2, ao3, bo0, bo1, bo2, bo3;                                               // index 2*i + q; BYTE offsets (host-checked < 4 GB)
#define G2_OFFS(j)                                                                                  \
    {                                                                                               \
        const int i_ = (j) >> 1, q_ = (j) & 1;                                                      \
        int64_t ar = m0 + i_ * 128 + q_ * 64 + r;                                                   \
        int64_t br = n0 + ((r >> 5) + 2 * i_) * 64 + q_ * 32 + (r & 31);                            \
        if (ar > M - 1) ar = M - 1;                                                                 \
        if (br > N - 1) br = N - 1;                                                                 \
        ao##j = (uint32_t)((ar * lda + schunk * 8) * 2); bo##j = (uint32_t)((br * ldb + schunk * 8) * 2); \
    }
    G2_OFFS(0) G2_OFFS(1) G2_OFFS(2) G2_OFFS(3)
#undef G2_OFFS
    unsigned short* const ldsw = lds_raw + wave * 8 * G2_BK;                                        // this wave's rows of round 0 inside a slot (scalar)
    const char* const Ab = reinterpret_cast<const char*>(A);
    const char* const Bb = reinterpret_cast<const char*>(B);
    // scalar base (operand + K offset) + 32-bit lane offset: global_load_lds_dwordx4 v_off, s[base:base+1] -- no VALU per DMA
#define G2_DMA(base_, off_, buf_, slot_, i_) MAED_LDS_DMA16(base_, off_, ldsw + ((buf_) * 4 + (slot_)) * G2_SLOT + (i_) * 64 * G2_BK)
    // issue half-tile `slot_` of K tile kt_ into buffer buf_ (kt_ is always a real K tile: the tail issues nothing)
#define G2_ISSUE(buf_, slot_, kt_) G2_ISSUE1(buf_, slot_, kt_, 0) G2_ISSUE1(buf_, slot_, kt_, 1)
    // round i_ (0/1) of half-tile slot_ of K tile kt_ into buffer buf_: ONE LDS-DMA per thread
#define G2_ISSUE1(buf_, slot_, kt_, i_)                                                             \
    if (!(ablate & 2)) {                                                                            \
        const char* const ak__ = Ab + (int64_t)(kt_) * (G2_BK * 2);                                 \
        const char* const bk__ = Bb + (int64_t)(kt_) * (G2_BK * 2);                                 \
        if ((slot_) == G2_A0) { if ((i_) == 0) G2_DMA(ak__, ao0, buf_, G2_A0, 0); else G2_DMA(ak__, ao2, buf_, G2_A0, 1); } \
        if ((slot_) == G2_A1) { if ((i_) == 0) G2_DMA(ak__, ao1, buf_, G2_A1, 0); else G2_DMA(ak__, ao3, buf_, G2_A1, 1); } \
        if ((slot_) == G2_B0) { if ((i_) == 0) G2_DMA(bk__, bo0, buf_, G2_B0, 0); else G2_DMA(bk__, bo2, buf_, G2_B0, 1); } \
        if ((slot_) == G2_B1) { if ((i_) == 0) G2_DMA(bk__, bo1, buf_, G2_B1, 0); else G2_DMA(bk__, bo3, buf_, G2_B1, 1); } \
    }

    // ---- fragments: A rows wr*64 + rt*32 + l31 of slot A[qm], B rows wc*32 + l31 of slot B[qn]; chunk (2*kk + hi) ^ fsw.
    //      One base per (operand, kk, buffer): everything else -- slot, rt -- is a compile-time offset < 64 KB (the ds_read immediate)
    const int fsw = (l31 >> 1) & 7;
    const char* const ldsb = reinterpret_cast<const char*>(lds_raw);
    const char* const fa0 = ldsb + (wr * 64 + l31) * (G2_BK * 2) + ((0 + hi) ^ fsw) * 16;
    const char* const fa1 = ldsb + (wr * 64 + l31) * (G2_BK * 2) + ((2 + hi) ^ fsw) * 16;
    const char* const fa2 = ldsb + (wr * 64 + l31) * (G2_BK * 2) + ((4 + hi) ^ fsw) * 16;
    const char* const fa3 = ldsb + (wr * 64 + l31) * (G2_BK * 2) + ((6 + hi) ^ fsw) * 16;
    const char* const fb0 = ldsb + (wc * 32 + l31) * (G2_BK * 2) + ((0 + hi) ^ fsw) * 16;
    const char* const fb1 = ldsb + (wc * 32 + l31) * (G2_BK * 2) + ((2 + hi) ^ fsw) * 16;
    const char* const fb2 = ldsb + (wc * 32 + l31) * (G2_BK * 2) + ((4 + hi) ^ fsw) * 16;
    const char* const fb3 = ldsb + (wc * 32 + l31) * (G2_BK * 2) + ((6 + hi) ^ fsw) * 16;
    bf16x8_t a00, a01, a02, a03, a10, a11, a12, a13;            // a[rt][kk]   (named scalars: never demoted to scratch)
    bf16x8_t b00, b01, b02, b03, b10, b11, b12, b13;            // b[qn][kk]
    f32x16_t c000, c001, c010, c011, c100, c101, c110, c111;    // c[qm][rt][qn]
#pragma unroll
    for (int x = 0; x < 16; ++x) { c000[x] = 0.f; c001[x] = 0.f; c010[x] = 0.f; c011[x] = 0.f; c100[x] = 0.f; c101[x] = 0.f; c110[x] = 0.f; c111[x] = 0.f; }
#define G2_FRAG(base_, buf_, slot_, rowoff_) (*reinterpret_cast<const bf16x8_t*>((base_) + (((buf_) * 4 + (slot_)) * G2_SLOT + (rowoff_) * G2_BK) * 2))
#define G2_READ_A(buf_, slot_)                                                                                              \
    a00 = G2_FRAG(fa0, buf_, slot_, 0); a01 = G2_FRAG(fa1, buf_, slot_, 0); a02 = G2_FRAG(fa2, buf_, slot_, 0); a03 = G2_FRAG(fa3, buf_, slot_, 0); \
    a10 = G2_FRAG(fa0, buf_, slot_, 32); a11 = G2_FRAG(fa1, buf_, slot_, 32); a12 = G2_FRAG(fa2, buf_, slot_, 32); a13 = G2_FRAG(fa3, buf_, slot_, 32);
#define G2_READ_B0(buf_) b00 = G2_FRAG(fb0, buf_, G2_B0, 0); b01 = G2_FRAG(fb1, buf_, G2_B0, 0); b02 = G2_FRAG(fb2, buf_, G2_B0, 0); b03 = G2_FRAG(fb3, buf_, G2_B0, 0);
#define G2_READ_B1(buf_) b10 = G2_FRAG(fb0, buf_, G2_B1, 0); b11 = G2_FRAG(fb1, buf_, G2_B1, 0); b12 = G2_FRAG(fb2, buf_, G2_B1, 0); b13 = G2_FRAG(fb3, buf_, G2_B1, 0);
    // transposed tiles (first operand = weight rows): a lane owns one output ROW and 4 consecutive columns per register group.
    // One phase: fragment reads ; barrier ; fragments landed ; 8 MFMAs at raised priority with the phase's two LDS-DMA instructions issued
    // in their shadow (an LDS-DMA costs the wave 100-185 issue cycles next to fragment reads, ~60 between MFMAs: MI355X_MICROARCH.md) ;
    // counted wait ; barrier.
#define G2_PHASE(READS_, COND_, buf_, slot_, kt_, WAIT_, c0_, c1_, bq_)                              \
    if (!(ablate & 4)) { READS_ }                                                                   \
    __builtin_amdgcn_s_barrier();                                                                   \
    MAED_WAIT_LGKMCNT0();                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    __builtin_amdgcn_s_setprio(1);                                                                  \
    if (!(ablate & 4)) {                                                                            \
        c0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##0, a00, c0_, 0, 0, 0);                   \
        c1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##0, a10, c1_, 0, 0, 0);                   \
    }                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    if (COND_) G2_ISSUE1(buf_, slot_, kt_, 0)                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    if (!(ablate & 4)) {                                                                            \
        c0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##1, a01, c0_, 0, 0, 0);                   \
        c1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##1, a11, c1_, 0, 0, 0);                   \
        c0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##2, a02, c0_, 0, 0, 0);                   \
    }                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    if (COND_) G2_ISSUE1(buf_, slot_, kt_, 1)                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    if (!(ablate & 4)) {                                                                            \
        c1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##2, a12, c1_, 0, 0, 0);                   \
        c0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##3, a03, c0_, 0, 0, 0);                   \
        c1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##3, a13, c1_, 0, 0, 0);                   \
    }                                                                                               \
    __builtin_amdgcn_s_setprio(0);                                                                  \
    WAIT_;                                                                                          \
    __builtin_amdgcn_s_barrier();
    // K tile t_ in buffer b_ (o_ = the other buffer); h1_ / h2_: K tile t+1 / t+2 exists (wave-uniform).  Issue order = consumption
    // order, six to seven phases ahead of the read: phase q of tile t issues  q1: A1(t+1)  q2: A0(t+2)  q3: B0(t+2)  q4: B1(t+2).
    // Waits sit at the END of a phase (behind its issue): vmcnt(8) = the half-tile issued four phases ago has landed; it is read two
    // phases later at the earliest (the other wave group runs one barrier behind), and a slot is re-targeted one phase after its last
    // read at the earliest (the issue sits behind the phase's first barrier).  The tail drains with the exact counts.
#define G2_KTILE_STEADY(b_, o_, t_)                                                                                                           \
    {                                                                                                                                         \
        G2_PHASE(G2_READ_A(b_, G2_A0) G2_READ_B0(b_), true, o_, G2_A1, (t_) + 1, MAED_WAIT_VMCNT(8), c000, c010, b0)                          \
        G2_PHASE(G2_READ_B1(b_), true, b_, G2_A0, (t_) + 2, MAED_WAIT_VMCNT(8), c001, c011, b1)                                               \
        G2_PHASE(G2_READ_A(b_, G2_A1), true, b_, G2_B0, (t_) + 2, MAED_WAIT_VMCNT(8), c101, c111, b1)                                         \
        G2_PHASE(, true, b_, G2_B1, (t_) + 2, MAED_WAIT_VMCNT(8), c100, c110, b0)                                                             \
    }
#define G2_KTILE(b_, o_, t_)                                                                                                                  \
    {                                                                                                                                         \
        const bool h1_ = (t_) + 1 < nkt, h2_ = (t_) + 2 < nkt;                                                                                \
        G2_PHASE(G2_READ_A(b_, G2_A0) G2_READ_B0(b_), h1_, o_, G2_A1, (t_) + 1, if (h1_) MAED_WAIT_VMCNT(8); else MAED_WAIT_VMCNT(0), c000, c010, b0) \
        G2_PHASE(G2_READ_B1(b_), h2_, b_, G2_A0, (t_) + 2, if (h2_) MAED_WAIT_VMCNT(8); else if (h1_) MAED_WAIT_VMCNT(6), c001, c011, b1)       \
        G2_PHASE(G2_READ_A(b_, G2_A1), h2_, b_, G2_B0, (t_) + 2, if (h2_) MAED_WAIT_VMCNT(8); else if (h1_) MAED_WAIT_VMCNT(4), c101, c111, b1) \
        G2_PHASE(, h2_, b_, G2_B1, (t_) + 2, if (h2_) MAED_WAIT_VMCNT(8); else if (h1_) MAED_WAIT_VMCNT(2), c100, c110, b0)                    \
    }

    // ---- prologue: K tile 0 whole and A0, B0, B1 of K tile 1 (the issue order of the steady state); A0, B0, B1 of tile 0 must have landed
    G2_ISSUE(0, G2_A0, 0) G2_ISSUE(0, G2_B0, 0) G2_ISSUE(0, G2_B1, 0) G2_ISSUE(0, G2_A1, 0) G2_ISSUE(1, G2_A0, 1) G2_ISSUE(1, G2_B0, 1) G2_ISSUE(1, G2_B1, 1)
    MAED_WAIT_VMCNT(8);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();        // waves 4-7 run one barrier behind (wave-uniform branch)
    int t = 0;
    for (; t + 3 < nkt; t += 2) {                     // steady state, two tiles per trip (static buffer index): tiles t, t+1 <= nkt-3
        G2_KTILE_STEADY(0, 1, t)
        G2_KTILE_STEADY(1, 0, t + 1)
    }
    for (; t < nkt; t += 2) {                         // the last two or three tiles: issue and wait by what is left
        G2_KTILE(0, 1, t)
        if (t + 1 < nkt) G2_KTILE(1, 0, t + 1)
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();        // re-align the two groups: nobody reads operand tiles past this point

    // ---- epilogue: per (qm, rt) a 32 x 64 piece of the wave's tile through its private LDS staging area
    const bool vec_ok = (e.ldo % 8 == 0) && (e.ldaux % 8 == 0);
    float* stg = reinterpret_cast<float*>(lds_raw) + wave * 32 * GL_ST;
    const int rr = lane >> 3, cc = (lane & 7) * 8;
#define G2_STORE_PIECE(accA_, accB_, qm_, rt_)                                                                              \
    __syncthreads();                                                                                                        \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                         \
        *reinterpret_cast<float4*>(stg + l31 * GL_ST + 8 * g + 4 * hi) = make_float4(accA_[4 * g], accA_[4 * g + 1], accA_[4 * g + 2], accA_[4 * g + 3]);      \
        *reinterpret_cast<float4*>(stg + l31 * GL_ST + 32 + 8 * g + 4 * hi) = make_float4(accB_[4 * g], accB_[4 * g + 1], accB_[4 * g + 2], accB_[4 * g + 3]); \
    }                                                                                                                       \
    __syncthreads();                                                                                                        \
    _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                                                      \
        const int lr = ps * 8 + rr;                                                                                         \
        const int64_t row = m0 + wr * 128 + (qm_) * 64 + (rt_) * 32 + lr, c0 = n0 + wc * 64 + cc;                           \
        float v8[8];                                                                                                        \
        ld8(stg + lr * GL_ST + cc, v8);                                                                                     \
        if (row < M && c0 < N && !(ablate & 1)) epilogue_store8<EPI, bf16>(e, row, c0, N, v8, vec_ok);                      \
    }
    G2_STORE_PIECE(c000, c001, 0, 0)
    G2_STORE_PIECE(c010, c011, 0, 1)
    G2_STORE_PIECE(c100, c101, 1, 0)
    G2_STORE_PIECE(c110, c111, 1, 1)
#undef G2_STORE_PIECE
}

template <int EPI>
static void launch_256(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const EpiArgs& e, hipStream_t s) {
    const int tm = (int)((M + G2_T - 1) / G2_T), tn = (int)((N + G2_T - 1) / G2_T);
#ifdef MAED_GEMM_ABLATE
    hipLaunchKernelGGL((gemm_nt_256_bf16_kernel<EPI>), dim3((unsigned)(tm * tn)), dim3(512), 0, s, (const bf16*)A, lda, (const bf16*)B, ldb, M, N, K, tn, e, maed_opt(MAED_OPT_ABLATE));
#else
    hipLaunchKernelGGL((gemm_nt_256_bf16_kernel<EPI>), dim3((unsigned)(tm * tn)), dim3(512), 0, s, (const bf16*)A, lda, (const bf16*)B, ldb, M, N, K, tn, e);
#endif
}

// called by maed_gemm_nt's dispatcher (gemm.hip); returns false for epilogues this kernel does not carry (fp32 atomics: split-K)
bool maed_gemm_nt_256_launch(int epilogue, const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const EpiArgs& e,
                             hipStream_t s) {
    switch (epilogue) {
        case MAED_EPI_STORE: launch_256<MAED_EPI_STORE>(A, lda, B, ldb, M, N, K, e, s); return true;
        case MAED_EPI_GELU: launch_256<MAED_EPI_GELU>(A, lda, B, ldb, M, N, K, e, s); return true;
        case MAED_EPI_RESID_F32: launch_256<MAED_EPI_RESID_F32>(A, lda, B, ldb, M, N, K, e, s); return true;
        case MAED_EPI_MUL_DGELU: launch_256<MAED_EPI_MUL_DGELU>(A, lda, B, ldb, M, N, K, e, s); return true;
        case MAED_EPI_STORE_F32: launch_256<MAED_EPI_STORE_F32>(A, lda, B, ldb, M, N, K, e, s); return true;
        case MAED_EPI_TANH: launch_256<MAED_EPI_TANH>(A, lda, B, ldb, M, N, K, e, s); return true;
        case MAED_EPI_ADD: launch_256<MAED_EPI_ADD>(A, lda, B, ldb, M, N, K, e, s); return true;
        default: return false;
    }
}
