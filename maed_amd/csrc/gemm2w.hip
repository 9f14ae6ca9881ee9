// nn.Linear family on 256 x 128 tiles, FOUR waves per workgroup, TWO workgroups per CU (round 6; bf16, fp32 accumulation).
// vision_transformer.py:98-111 (Mlp fc1 / fc2), :124-128,147 (qkv), :176 (proj) and their input gradients through autograd.
//
// Why this geometry (profiles/r06_sk_micro.txt).  The 128 x 128 kernel (gemm.hip, four workgroups per CU) is bound by L2 -> LDS operand
// traffic; the 256 x 256 pipeline (gemm256.hip) halves that traffic but runs ONE workgroup per CU, and on gfx950 a lone workgroup cannot hide its own epilogue:
// loads and stores retire in order through one counter, so the stores of a finished tile stand in front of the next tile's copies (the persistent form of round
// 6, gemm_sk.hip, measured exactly that).  Overlap of one tile's epilogue / prologue with another tile's MFMAs has to come from a SECOND workgroup on the CU.
// 256 x 128 tiles on 4 waves keep gemm256's per-wave shape -- 128 x 64 outputs, 8 accumulator tiles, 12 fragment reads per 16 MFMAs -- at 0.75x the operand bytes
// of 128 x 128, and fit twice: <= 256 VGPRs (one wave per SIMD and workgroup, two workgroups), 72 KB of LDS.
//
// Pipeline.  K steps of 32 (a stage = A 256 x 32 + B 128 x 32 bf16 = 24 KB = six LDS-DMA instructions per thread), a ring of THREE stages, copies two steps
// ahead, ONE barrier per step:   wait vmcnt(6) (stage k landed, k+1 in flight) -> barrier (every wave's part landed; everybody is done with stage k-1) -> issue
// stage k+2 into the slot stage k-1 used -> 12 fragment reads -> 16 MFMAs.  Raw s_barrier + counted vmcnt (the copies are inline assembly the compiler does not
// count).  LDS image: [row][32 k] bf16 = 64-byte rows, 16-byte chunk c of row r at slot c ^ ((r >> 2) & 3) -- applied on the copy's SOURCE address (an LDS-DMA
// writes base + lane * 16) and on the ds_read_b128 fragment reads: the 16 lanes of a read group hit 16 different bank quads.
// Epilogue: the fused epilogues of gemm_epilogue.cuh through the LDS shuffle of gemm256.hip (the ring is free by then): full 128 / 256-byte row segments.
#include "common.cuh"
#include "gemm_epilogue.cuh"

#define W2_BM 256
#define W2_BN 128
#define W2_BK 32
#define W2_STAGE ((W2_BM + W2_BN) * W2_BK)      // elements per stage: 12288 (24 KB)
// NST: ring stages (3: two workgroups per CU, copies two steps ahead; 2: 48 KB, THREE workgroups per CU, copies one step ahead).
// INTER: the six copy instructions of a step issued between its MFMAs (their issue cost -- 60-180 cycles each -- in the matrix pipe's shadow) instead of in front.
template <int EPI, int NST, bool INTER>
__global__ __launch_bounds__(256, (NST == 2 ? 3 : 2)) void gemm_nt_2w_bf16_kernel(const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ B, int64_t ldb,
                                                                  int64_t M, int64_t N, int64_t K, int tiles_n, EpiArgs e) {
    __shared__ __attribute__((aligned(1024))) unsigned short lds_raw[NST * W2_STAGE];          // 72 / 48 KB; the epilogue re-uses it
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(id / tiles_n) * W2_BM, n0 = (int64_t)(id % tiles_n) * W2_BN;
    const int nk = (int)(K / W2_BK);                                                            // >= 2 (host-checked)

    // ---- copies: a stage is 384 rows x 64 B (A rows 0..255, then B rows 0..127) = 24 groups of 16 rows = 24 wave instructions of 1 KB; wave w issues groups
    //      4 j + w, j = 0..5 (j <= 3: A, j = 4, 5: B); lane l -> row 16 g + (l >> 2), chunk SLOT l & 3; the chunk that belongs there undoes the swizzle
    uint32_t o0, o1, o2, o3, o4, o5;                                                            // BYTE offsets (host-checked < 4 GB)
    {
        const uint32_t lda2 = (uint32_t)lda * 2u, ldb2 = (uint32_t)ldb * 2u;
        const int Mm1 = (int)M - 1, Nm1 = (int)N - 1;
#define W2_OFF(j)                                                                                   \
        {                                                                                           \
            const int g_ = 4 * (j) + wave, r_ = 16 * (g_ & 15) + (lane >> 2);                       \
            const uint32_t ch_ = (uint32_t)((lane & 3) ^ ((r_ >> 2) & 3)) * 16u;                    \
            if ((j) < 4) { int ar = (int)m0 + r_; ar = ar > Mm1 ? Mm1 : ar; o##j = (uint32_t)ar * lda2 + ch_; } \
            else { int br = (int)n0 + r_; br = br > Nm1 ? Nm1 : br; o##j = (uint32_t)br * ldb2 + ch_; }         \
        }
        W2_OFF(0) W2_OFF(1) W2_OFF(2) W2_OFF(3) W2_OFF(4) W2_OFF(5)
#undef W2_OFF
    }
    // (for j >= 4 the group index 4 j + w runs 16..23: B row = 16 (g - 16) + ... = 16 (g & 15) + ... because g & 15 = g - 16 there; A: g & 15 = g)
    unsigned short* const ldsw = lds_raw + wave * 16 * W2_BK;                                    // this wave's 16 rows of group w inside a stage (scalar)
    const char* const Ab = reinterpret_cast<const char*>(A);
    const char* const Bb = reinterpret_cast<const char*>(B);
#define W2_DMA(base_, off_, st_, j_) MAED_LDS_DMA16(base_, off_, ldsw + (st_) * W2_STAGE + (j_) * 4 * 16 * W2_BK)
#define W2_ISSUE(st_, ks_)                                                                          \
    {                                                                                               \
        const char* const ak__ = Ab + (int64_t)(ks_) * (W2_BK * 2);                                 \
        const char* const bk__ = Bb + (int64_t)(ks_) * (W2_BK * 2);                                 \
        W2_DMA(ak__, o0, st_, 0); W2_DMA(ak__, o1, st_, 1); W2_DMA(ak__, o2, st_, 2); W2_DMA(ak__, o3, st_, 3); \
        W2_DMA(bk__, o4, st_, 4); W2_DMA(bk__, o5, st_, 5);                                         \
    }

    // ---- fragments: A rows wr * 128 + blk * 32 + l31 (blk 0..3), B rows 256 + wc * 64 + qn * 32 + l31 of the stage; chunk (2 kk + hi) ^ ((l31 >> 2) & 3)
    const int fsw = (l31 >> 2) & 3;
    const char* const ldsb = reinterpret_cast<const char*>(lds_raw);
    const char* const fa0 = ldsb + (wr * 128 + l31) * (W2_BK * 2) + ((0 + hi) ^ fsw) * 16;         // kk = 0
    const char* const fa1 = ldsb + (wr * 128 + l31) * (W2_BK * 2) + ((2 + hi) ^ fsw) * 16;         // kk = 1
    const char* const fb0 = ldsb + (W2_BM + wc * 64 + l31) * (W2_BK * 2) + ((0 + hi) ^ fsw) * 16;
    const char* const fb1 = ldsb + (W2_BM + wc * 64 + l31) * (W2_BK * 2) + ((2 + hi) ^ fsw) * 16;
    bf16x8_t a00, a01, a10, a11, a20, a21, a30, a31;            // a[blk][kk]
    bf16x8_t b00, b01, b10, b11;                                // b[qn][kk]
    f32x16_t c00, c01, c10, c11, c20, c21, c30, c31;            // c[blk][qn]
#pragma unroll
    for (int x = 0; x < 16; ++x) { c00[x] = 0.f; c01[x] = 0.f; c10[x] = 0.f; c11[x] = 0.f; c20[x] = 0.f; c21[x] = 0.f; c30[x] = 0.f; c31[x] = 0.f; }
#define W2_FRAG(base_, st_, rowoff_) (*reinterpret_cast<const bf16x8_t*>((base_) + ((st_) * W2_STAGE + (rowoff_) * W2_BK) * 2))
    // one K step on stage st_: 12 fragment reads, 16 MFMAs (transposed tiles: first operand = weight rows, a lane owns one output ROW)
#define W2_MF(c_, b_, a_) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_, a_, c_, 0, 0, 0);
#define W2_SB() __builtin_amdgcn_sched_barrier(0);
    // S0 .. S5: the statements issued between the MFMA groups (copies of a later stage, or nothing)
#define W2_STEP(st_, S0, S1, S2, S3, S4, S5)                                                        \
    a00 = W2_FRAG(fa0, st_, 0); a01 = W2_FRAG(fa1, st_, 0); b00 = W2_FRAG(fb0, st_, 0); b01 = W2_FRAG(fb1, st_, 0); \
    a10 = W2_FRAG(fa0, st_, 32); a11 = W2_FRAG(fa1, st_, 32); b10 = W2_FRAG(fb0, st_, 32); b11 = W2_FRAG(fb1, st_, 32); \
    a20 = W2_FRAG(fa0, st_, 64); a21 = W2_FRAG(fa1, st_, 64); a30 = W2_FRAG(fa0, st_, 96); a31 = W2_FRAG(fa1, st_, 96); \
    __builtin_amdgcn_s_setprio(1);                                                                  \
    W2_MF(c00, b00, a00) W2_MF(c01, b10, a00) W2_SB() S0 W2_SB()                                    \
    W2_MF(c10, b00, a10) W2_MF(c11, b10, a10) W2_SB() S1 W2_SB()                                    \
    W2_MF(c00, b01, a01) W2_MF(c01, b11, a01) W2_SB() S2 W2_SB()                                    \
    W2_MF(c10, b01, a11) W2_MF(c11, b11, a11) W2_MF(c20, b00, a20) W2_SB() S3 W2_SB()               \
    W2_MF(c21, b10, a20) W2_MF(c30, b00, a30) W2_MF(c31, b10, a30) W2_SB() S4 W2_SB()               \
    W2_MF(c20, b01, a21) W2_MF(c21, b11, a21) W2_SB() S5 W2_SB()                                    \
    W2_MF(c30, b01, a31) W2_MF(c31, b11, a31)                                                       \
    __builtin_amdgcn_s_setprio(0);
    // one iteration: stage ks_ has landed (its six copies are the oldest outstanding) -> barrier -> refill the slot the previous stage used -> compute
#define W2_ITER(st_, nst_, ks_)                                                                     \
    if (NST == 3 && (ks_) + 1 < nk) MAED_WAIT_VMCNT(6); else MAED_WAIT_VMCNT0();                    \
    __builtin_amdgcn_s_barrier();                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    {                                                                                               \
        const bool more__ = (ks_) + (NST - 1) < nk;                                                 \
        const char* const ak__ = Ab + (int64_t)((ks_) + (NST - 1)) * (W2_BK * 2);                   \
        const char* const bk__ = Bb + (int64_t)((ks_) + (NST - 1)) * (W2_BK * 2);                   \
        if (INTER) {       /* (ONE body, the copies predicated by a wave-uniform branch each: two bodies that both write the accumulators are merged with copies) */ \
            W2_STEP(st_, if (more__) W2_DMA(ak__, o0, nst_, 0);, if (more__) W2_DMA(ak__, o1, nst_, 1);, if (more__) W2_DMA(ak__, o2, nst_, 2);,                      \
                    if (more__) W2_DMA(ak__, o3, nst_, 3);, if (more__) W2_DMA(bk__, o4, nst_, 4);, if (more__) W2_DMA(bk__, o5, nst_, 5);)                           \
        } else {                                                                                    \
            if (more__) W2_ISSUE(nst_, (ks_) + (NST - 1))                                           \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            W2_STEP(st_, , , , , , )                                                                \
        }                                                                                           \
    }

    W2_ISSUE(0, 0)
    if constexpr (NST == 3) {
        W2_ISSUE(1, 1)
        int ks = 0;
        for (; ks + 2 < nk; ks += 3) {            // three steps per trip: static stage indices
            W2_ITER(0, 2, ks)
            W2_ITER(1, 0, ks + 1)
            W2_ITER(2, 1, ks + 2)
        }
        if (ks < nk) { W2_ITER(0, 2, ks) }
        if (ks + 1 < nk) { W2_ITER(1, 0, ks + 1) }
    } else {
        int ks = 0;
        for (; ks + 1 < nk; ks += 2) {
            W2_ITER(0, 1, ks)
            W2_ITER(1, 0, ks + 1)
        }
        if (ks < nk) { W2_ITER(0, 1, ks) }
    }

    // ---- epilogue: per block a 32 x 64 piece of the wave's tile through its private LDS staging area (a lane owns one output row in the transposed accumulators;
    //      each wave parks 32 x 64 fp32 and stores full row segments: 8 lanes cover 64 columns)
    const bool vec_ok = (e.ldo % 8 == 0) && (e.ldaux % 8 == 0);
    float* stg = reinterpret_cast<float*>(lds_raw) + wave * 32 * GL_ST;
    const int rr = lane >> 3, cc = (lane & 7) * 8;
#define W2_STORE_PIECE(accA_, accB_, blk_)                                                                                  \
    __syncthreads();                                                                                                        \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                         \
        *reinterpret_cast<float4*>(stg + l31 * GL_ST + 8 * g + 4 * hi) = make_float4(accA_[4 * g], accA_[4 * g + 1], accA_[4 * g + 2], accA_[4 * g + 3]);      \
        *reinterpret_cast<float4*>(stg + l31 * GL_ST + 32 + 8 * g + 4 * hi) = make_float4(accB_[4 * g], accB_[4 * g + 1], accB_[4 * g + 2], accB_[4 * g + 3]); \
    }                                                                                                                       \
    __syncthreads();                                                                                                        \
    _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                                                      \
        const int lr = ps * 8 + rr;                                                                                         \
        const int64_t row = m0 + wr * 128 + (blk_) * 32 + lr, c0 = n0 + wc * 64 + cc;                                       \
        float v8[8];                                                                                                        \
        ld8(stg + lr * GL_ST + cc, v8);                                                                                     \
        if (row < M && c0 < N) epilogue_store8<EPI, bf16>(e, row, c0, N, v8, vec_ok);                                       \
    }
    W2_STORE_PIECE(c00, c01, 0)
    W2_STORE_PIECE(c10, c11, 1)
    W2_STORE_PIECE(c20, c21, 2)
    W2_STORE_PIECE(c30, c31, 3)
#undef W2_STORE_PIECE
}

template <int EPI>
static void launch_2w(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const EpiArgs& e, hipStream_t s) {
    const int tm = (int)((M + W2_BM - 1) / W2_BM), tn = (int)((N + W2_BN - 1) / W2_BN);
    const dim3 grid((unsigned)(tm * tn));
    // variant (sweep knob MAED_OPT_SK_GRID while this kernel is being tuned): 0 = three stages, copies between the MFMAs; 1 = three stages, copies in front;
    // 2 = two stages (three workgroups per CU), copies between; 3 = two stages, copies in front
    switch (maed_opt(MAED_OPT_SK_GRID)) {
        case 1: hipLaunchKernelGGL((gemm_nt_2w_bf16_kernel<EPI, 3, false>), grid, dim3(256), 0, s, (const bf16*)A, lda, (const bf16*)B, ldb, M, N, K, tn, e); break;
        case 2: hipLaunchKernelGGL((gemm_nt_2w_bf16_kernel<EPI, 2, true>), grid, dim3(256), 0, s, (const bf16*)A, lda, (const bf16*)B, ldb, M, N, K, tn, e); break;
        case 3: hipLaunchKernelGGL((gemm_nt_2w_bf16_kernel<EPI, 2, false>), grid, dim3(256), 0, s, (const bf16*)A, lda, (const bf16*)B, ldb, M, N, K, tn, e); break;
        default: hipLaunchKernelGGL((gemm_nt_2w_bf16_kernel<EPI, 3, true>), grid, dim3(256), 0, s, (const bf16*)A, lda, (const bf16*)B, ldb, M, N, K, tn, e); break;
    }
}

// called by maed_gemm_nt's dispatcher (gemm.hip); returns false for epilogues this kernel does not carry
bool maed_gemm_nt_2w_launch(int epilogue, const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const EpiArgs& e,
                            hipStream_t s) {
    switch (epilogue) {
        case MAED_EPI_STORE: launch_2w<MAED_EPI_STORE>(A, lda, B, ldb, M, N, K, e, s); return true;
        case MAED_EPI_GELU: launch_2w<MAED_EPI_GELU>(A, lda, B, ldb, M, N, K, e, s); return true;
        case MAED_EPI_RESID_F32: launch_2w<MAED_EPI_RESID_F32>(A, lda, B, ldb, M, N, K, e, s); return true;
        case MAED_EPI_MUL_DGELU: launch_2w<MAED_EPI_MUL_DGELU>(A, lda, B, ldb, M, N, K, e, s); return true;
        case MAED_EPI_STORE_F32: launch_2w<MAED_EPI_STORE_F32>(A, lda, B, ldb, M, N, K, e, s); return true;
        default: return false;
    }
}
