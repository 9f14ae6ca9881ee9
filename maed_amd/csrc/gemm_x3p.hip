// Split-bf16 ("bf16x3") NT GEMM on operands that are ALREADY stored as two bf16 planes (round 5):  out = epi((Ah + Al)[M,K] (Bh + Bl)[N,K]^T)
// with the three partial products  Ah Bl + Al Bh + Ah Bh  per K step, fp32 accumulators -- the arithmetic of gemm_x3.hip's gemm_nt_x3_kernel<*, 2> bit for bit
// (same K order, same product order), when the planes are that kernel's split of the fp32 operand: hi = bf16(x) (round to nearest even), lo = bf16(x - hi).
//
// Why a second kernel: gemm_nt_x3_kernel reads fp32 from HBM into registers, splits on the VALU and WRITES the planes into LDS (32 KB per 24 MFMAs and wave): its
// LDS port carries the plane writes AND the fragment reads, which together take as long as the MFMAs, behind two barriers per K tile.  With the
// planes in memory -- the same 4 bytes per element -- a K tile is 32 LDS-DMA copies of 1 KB (global_load_lds_dwordx4: no registers, no VALU, no ds_write), a ring of
// stages keeps copies in flight under the MFMAs, and there is ONE barrier per tile.  The hi plane is at the same time the bf16 twin the bf16 backward reads
// (ops.py "twin mode"): an activation stored as planes costs 4 bytes per element where fp32 + twin cost 6.
//
// LDS stage (32 KB): [Ah | Al | Bh | Bl], each 128 rows x 32 k (64-byte rows, unpadded); the 16-byte chunk c of row r lives in slot c ^ ((r >> 2) & 3): the 16 lanes
// of a ds_read_b128 pass (16 rows, one chunk index) then hit 16 different bank groups.  An LDS-DMA writes wave-uniform base + lane * 16, so the swizzle is applied to
// the SOURCE address: lane l of a copy fills row 16 j + (l >> 2), slot l & 3, with chunk (l & 3) ^ ((l >> 4) & 3) of that row -- 4 lanes per 64-byte row segment.
//   NST = 2: 64 KB, two workgroups per CU, next tile's copies under this tile's MFMAs (the layout of gemm.hip's gemm_nt_glds_bf16_kernel<*, 2>)
//   NST = 4: 128 KB, one workgroup per CU, copies three tiles ahead with counted waits (gemm_tn2.hip's ring)
#include "common.cuh"
#include "gemm_epilogue.cuh"
#include "gemm_x3.h"


namespace {

template <int N> __device__ __forceinline__ void xp_wait_vmcnt() {
    if constexpr (N == 0) { MAED_WAIT_VMCNT0(); } else if constexpr (N == 6) { MAED_WAIT_VMCNT(6); } else if constexpr (N == 8) { MAED_WAIT_VMCNT(8); }
    else if constexpr (N == 12) { MAED_WAIT_VMCNT(12); } else if constexpr (N == 16) { MAED_WAIT_VMCNT(16); } else { static_assert(N == 0, "count not listed"); }
}

// WM x WN waves, each RM x 2 accumulator tiles of 32 x 32: output tile TM = WM * RM * 32 rows by TN = WN * 64 columns.
//   <2, 2, 2>  128 x 128, four waves,  32 KB per stage: NST = 2 -> two workgroups per CU
//   <4, 2, 2>  256 x 128, eight waves, 48 KB per stage: NST = 3, one workgroup per CU (two waves per SIMD), 3/4 of the copies per MFMA
//   <2, 4, 4>  256 x 256, eight waves (128 x 64 each), 64 KB per stage: NST = 2, one workgroup per CU, half the copies per MFMA
//   <4, 2, 1>  128 x 128, eight waves (32 x 64 each), BK = 64 (full 128-byte lines per row and plane), 64 KB per stage: NST = 2, one workgroup per CU
template <int EPI, int NST, int WM, int WN, int RM, int XP_BK>
__global__ __launch_bounds__(WM * WN * 64, (NST * (WM * RM * 32 + WN * 64) * 2 * XP_BK * 2 <= 80 * 1024 ? 2 : 1))
void gemm_nt_x3p_kernel(const bf16* __restrict__ Ah, const bf16* __restrict__ Al, int64_t lda, const bf16* __restrict__ Bh, const bf16* __restrict__ Bl, int64_t ldb,
                        int64_t M, int64_t N, int64_t K, int tiles_n, EpiArgs e
#ifdef MAED_GEMM_ABLATE
                        , int ablate    // diagnostic build only (scripts/build_ablate.sh): 1 no stores, 2 no copies after the prologue, 4 no MFMA / fragment reads
#endif
                        ) {
#ifndef MAED_GEMM_ABLATE
    constexpr int ablate = 0;
#endif
    constexpr int NW = WM * WN, TM = WM * RM * 32, TN = WN * 64;
    constexpr int A_PLANE = TM * XP_BK, B_PLANE = TN * XP_BK, STAGE = 2 * A_PLANE + 2 * B_PLANE;       // elements
    constexpr int RPI = 512 / XP_BK, CPR = XP_BK / 8;                                                   // a copy instruction = 1 KB = RPI rows of one plane; chunks per row
    constexpr int SW_SH = XP_BK == 32 ? 2 : 1, SW_M = CPR - 1;                                          // slot of chunk c of row r: c ^ ((r >> SW_SH) & SW_M)
    constexpr int NINSTR = STAGE / 512, IPW = NINSTR / NW, A_INSTR = TM / RPI, B_INSTR = TN / RPI;
    static_assert(XP_BK == 32 || XP_BK == 64, "K tile");
    static_assert(EPI != MAED_EPI_ATOMIC_F32, "no split-K form");
    static_assert(NINSTR % NW == 0, "copies per wave");
    static_assert(NST * STAGE * 2 >= NW * 32 * GL_ST * 4, "the epilogue's staging area lives in the ring");
    MAED_DYN_SHARED(unsigned short, lds);                                   // NST stages
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN, l31 = lane & 31, hi = lane >> 5;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(id / tiles_n) * TM, n0 = (int64_t)(id % tiles_n) * TN;
    const int nt = (int)(K / XP_BK);

    // copies: instruction q of a stage fills rows 16 j .. 16 j + 15 of plane [Ah | Al | Bh | Bl]; wave w issues q = w IPW .. w IPW + IPW - 1
    uint32_t voff[IPW];
    const char* sbase[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int q = wave * IPW + i;
        const bool isb = q >= 2 * A_INSTR;
        const int qq = isb ? q - 2 * A_INSTR : q, per = isb ? B_INSTR : A_INSTR;
        const int lo = qq >= per, j = lo ? qq - per : qq;
        const int row = RPI * j + lane / CPR;
        const int chunk = (lane % CPR) ^ ((row >> SW_SH) & SW_M);
        const int64_t lim = isb ? N : M, r0 = isb ? n0 : m0;
        const int64_t gr = (r0 + row < lim) ? r0 + row : lim - 1;
        voff[i] = (uint32_t)((gr * (isb ? ldb : lda) + chunk * 8) * 2);
        sbase[i] = reinterpret_cast<const char*>(isb ? (lo ? Bl : Bh) : (lo ? Al : Ah));
    }
#define XP_ISSUE(t_) { const int t__ = (t_); unsigned short* const st__ = lds + (t__ % NST) * STAGE + wave * IPW * 512; const int64_t kb__ = (int64_t)t__ * (XP_BK * 2); \
        _Pragma("unroll") for (int i = 0; i < IPW; ++i) MAED_LDS_DMA16(sbase[i] + kb__, voff[i], st__ + i * 512); }

    f32x16_t acc[RM][2];
#pragma unroll
    for (int a = 0; a < RM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[a][0][r] = 0.f; acc[a][1][r] = 0.f; }
    const int fsw = (l31 >> SW_SH) & SW_M;
    const int a_off = (wr * RM * 32 + l31) * XP_BK, b_off = 2 * A_PLANE + (wc * 64 + l31) * XP_BK;

#pragma unroll
    for (int s = 0; s < NST - 1; ++s) if (s < nt) XP_ISSUE(s);
    for (int t = 0; t < nt; ++t) {
        // tile t has landed once at most the younger tiles' copies (IPW per wave and tile, issued in order) are outstanding
        const int ahead = (nt - 1 - t) < (NST - 2) ? (nt - 1 - t) : (NST - 2);
        if (NST >= 4 && ahead >= 2) xp_wait_vmcnt<(NST >= 4 ? 2 * IPW : 0)>(); else if (NST >= 3 && ahead == 1) xp_wait_vmcnt<(NST >= 3 ? IPW : 0)>(); else xp_wait_vmcnt<0>();
        __syncthreads();                     // ... for every wave; and every wave is done with tile t - 1: its stage is free
        if (t + NST - 1 < nt && !(ablate & 2)) XP_ISSUE(t + NST - 1);
        const unsigned short* const st = lds + (t % NST) * STAGE;
        if (ablate & 4) continue;
#pragma unroll
        for (int kk = 0; kk < XP_BK / 16; ++kk) {
            const int co = ((kk * 2 + hi) ^ fsw) * 8;
            bf16x8_t ah[RM], al[RM], bh[2], bl[2];
#pragma unroll
            for (int a = 0; a < RM; ++a) {
                ah[a] = *reinterpret_cast<const bf16x8_t*>(st + a_off + a * 32 * XP_BK + co);
                al[a] = *reinterpret_cast<const bf16x8_t*>(st + A_PLANE + a_off + a * 32 * XP_BK + co);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                bh[b] = *reinterpret_cast<const bf16x8_t*>(st + b_off + b * 32 * XP_BK + co);
                bl[b] = *reinterpret_cast<const bf16x8_t*>(st + B_PLANE + b_off + b * 32 * XP_BK + co);
            }
            // per accumulator: smallest partial products first, the leading hi x hi last (gemm_x3.hip's order); the accumulators hold the transposed tiles
#pragma unroll
            for (int a = 0; a < RM; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[b], ah[a], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < RM; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[b], al[a], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < RM; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[b], ah[a], acc[a][b], 0, 0, 0);
        }
    }
#undef XP_ISSUE

    // LDS-shuffled epilogue (gemm.hip / gemm_x3.hip): each wave parks a 32 x 64 piece in LDS and re-reads it with 8 lanes per row
    const bool vec_ok = (e.ldo % 8 == 0) && (e.ldaux % 8 == 0);
    float* stg = reinterpret_cast<float*>(lds) + wave * 32 * GL_ST;
    const int rr = lane >> 3, cc = (lane & 7) * 8;
#pragma unroll
    for (int a = 0; a < RM; ++a) {
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<float4*>(stg + l31 * GL_ST + 8 * g + 4 * hi) = make_float4(acc[a][0][4 * g], acc[a][0][4 * g + 1], acc[a][0][4 * g + 2], acc[a][0][4 * g + 3]);
            *reinterpret_cast<float4*>(stg + l31 * GL_ST + 32 + 8 * g + 4 * hi) = make_float4(acc[a][1][4 * g], acc[a][1][4 * g + 1], acc[a][1][4 * g + 2], acc[a][1][4 * g + 3]);
        }
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int lr = ps * 8 + rr;
            const int64_t row = m0 + wr * RM * 32 + a * 32 + lr, col0 = n0 + wc * 64 + cc;
            float v8[8];
            ld8(stg + lr * GL_ST + cc, v8);
            if (row < M && col0 < N && !(ablate & 1)) epilogue_store8<EPI, float>(e, row, col0, N, v8, vec_ok);
        }
    }
}

// fp32 -> (hi, lo) planes, 8 elements per thread and trip; `table` entries are independent tensors (one launch for a block's weights)
struct SplitTab { const float* src[8]; bf16* hi[8]; bf16* lo[8]; long long n8[8]; };
__global__ __launch_bounds__(256) void split_planes_kernel(SplitTab t) {
    const float* __restrict__ src = t.src[blockIdx.y];
    bf16* __restrict__ ph = t.hi[blockIdx.y];
    bf16* __restrict__ pl = t.lo[blockIdx.y];
    const long long n8 = t.n8[blockIdx.y];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        float v[8];
        ld8(src + i * 8, v);
        uint2 p0[2], p1[2];
        split4<2>(v[0], v[1], v[2], v[3], p0);
        split4<2>(v[4], v[5], v[6], v[7], p1);
        *reinterpret_cast<uint4*>(ph + i * 8) = make_uint4(p0[0].x, p0[0].y, p1[0].x, p1[0].y);
        *reinterpret_cast<uint4*>(pl + i * 8) = make_uint4(p0[1].x, p0[1].y, p1[1].x, p1[1].y);
    }
}

template <int EPI, int NST, int WM, int WN, int RM, int XP_BK = 32>
void launch_x3p(const bf16* Ah, const bf16* Al, int64_t lda, const bf16* Bh, const bf16* Bl, int64_t ldb, int64_t M, int64_t N, int64_t K, const EpiArgs& e,
                hipStream_t s) {
    constexpr int TM = WM * RM * 32, TN = WN * 64;
    constexpr size_t lds = (size_t)NST * (2 * TM + 2 * TN) * XP_BK * 2;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_nt_x3p_kernel<EPI, NST, WM, WN, RM, XP_BK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
    const int tm = (int)((M + TM - 1) / TM), tn = (int)((N + TN - 1) / TN);
#ifdef MAED_GEMM_ABLATE
    hipLaunchKernelGGL((gemm_nt_x3p_kernel<EPI, NST, WM, WN, RM, XP_BK>), dim3((unsigned)(tm * tn)), dim3(WM * WN * 64), lds, s, Ah, Al, lda, Bh, Bl, ldb, M, N, K, tn, e, maed_opt(MAED_OPT_ABLATE));
#else
    hipLaunchKernelGGL((gemm_nt_x3p_kernel<EPI, NST, WM, WN, RM, XP_BK>), dim3((unsigned)(tm * tn)), dim3(WM * WN * 64), lds, s, Ah, Al, lda, Bh, Bl, ldb, M, N, K, tn, e);
#endif
}

template <int EPI>
void launch_x3p_variant(int variant, const bf16* Ah, const bf16* Al, int64_t lda, const bf16* Bh, const bf16* Bl, int64_t ldb, int64_t M, int64_t N, int64_t K,
                        const EpiArgs& e, hipStream_t s) {
    switch (variant) {
        case 4: launch_x3p<EPI, 4, 2, 2, 2>(Ah, Al, lda, Bh, Bl, ldb, M, N, K, e, s); break;        // 128 x 128, four stages, one workgroup per CU
        case 5: launch_x3p<EPI, 3, 4, 2, 2>(Ah, Al, lda, Bh, Bl, ldb, M, N, K, e, s); break;        // 256 x 128, three stages
        case 6: launch_x3p<EPI, 2, 2, 4, 4>(Ah, Al, lda, Bh, Bl, ldb, M, N, K, e, s); break;        // 256 x 256, two stages
        case 7: if (K % 64 == 0) { launch_x3p<EPI, 2, 4, 2, 1, 64>(Ah, Al, lda, Bh, Bl, ldb, M, N, K, e, s); break; }     // 128 x 128, eight waves, K tiles of 64 (else: the default)
        default: launch_x3p<EPI, 2, 2, 2, 2>(Ah, Al, lda, Bh, Bl, ldb, M, N, K, e, s); break;       // 128 x 128, two stages, two workgroups per CU
    }
}

}  // namespace

bool maed_x3p_shape_ok(const void* Ah, const void* Al, int64_t lda, const void* Bh, const void* Bl, int64_t ldb, int64_t M, int64_t N, int64_t K) {
    return M > 0 && N > 0 && K >= 32 && K % 32 == 0 && lda % 8 == 0 && ldb % 8 == 0 && is_aligned(Ah, 16) && is_aligned(Al, 16) && is_aligned(Bh, 16)
           && is_aligned(Bl, 16) && (uint64_t)M * (uint64_t)lda * 2 < 0xfffffff0ull && (uint64_t)N * (uint64_t)ldb * 2 < 0xfffffff0ull;
}

// variant: 0 = what MAED_OPT_X3_PLANES names; 2 / 4 = 128 x 128 tiles with a ring of 2 / 4 stages, 5 = 256 x 128 tiles (3 stages), 6 = 256 x 256 tiles (2 stages), 7 = 128 x 128 tiles with K tiles of 64 on eight waves
int maed_gemm_nt_x3p_launch(int epilogue, int variant, const void* a_hi, const void* a_lo, int64_t lda, const void* b_hi, const void* b_lo, int64_t ldb, int64_t M,
                            int64_t N, int64_t K, const EpiArgs& e, hipStream_t s) {
    const bf16 *Ah = (const bf16*)a_hi, *Al = (const bf16*)a_lo, *Bh = (const bf16*)b_hi, *Bl = (const bf16*)b_lo;
    const int v = variant ? variant : maed_opt(MAED_OPT_X3_PLANES);
    switch (epilogue) {
        case MAED_EPI_STORE: launch_x3p_variant<MAED_EPI_STORE>(v, Ah, Al, lda, Bh, Bl, ldb, M, N, K, e, s); break;
        case MAED_EPI_GELU: launch_x3p_variant<MAED_EPI_GELU>(v, Ah, Al, lda, Bh, Bl, ldb, M, N, K, e, s); break;
        case MAED_EPI_RESID_F32: launch_x3p_variant<MAED_EPI_RESID_F32>(v, Ah, Al, lda, Bh, Bl, ldb, M, N, K, e, s); break;
        default: maed_set_error("gemm_nt_planes: epilogue %d is not built for plane operands", epilogue); return MAED_ERR_ARG;
    }
    return MAED_OK;
}

int maed_split_planes_launch(int count, const float* const* src, void* const* hi, void* const* lo, const int64_t* n, hipStream_t s) {
    SplitTab t{};
    long long most = 0;
    for (int i = 0; i < count; ++i) { t.src[i] = src[i]; t.hi[i] = (bf16*)hi[i]; t.lo[i] = (bf16*)lo[i]; t.n8[i] = n[i] / 8; if (t.n8[i] > most) most = t.n8[i]; }
    if (count == 0 || most == 0) return MAED_OK;
    long long bx = (most + 255) / 256;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)bx, (unsigned)count), dim3(256), 0, s, t);
    return MAED_OK;
}

// ---- C-ABI (include/maed_hip.h) -----------------------------------------------------------------------------------------------------------------------------
extern "C" int maed_gemm_nt_planes(const void* a_hi, const void* a_lo, int64_t lda, const void* b_hi, const void* b_lo, int64_t ldb, int64_t M, int64_t N, int64_t K,
                                   int epilogue, const float* bias, void* out, int64_t ldo, void* out2_bf16, const void* aux, int64_t ldaux, void* out_hi,
                                   void* out_lo, int variant, void* stream) {
    MAED_CHECK_ARG(a_hi && a_lo && b_hi && b_lo, MAED_ERR_ARG, "gemm_nt_planes: null operand plane");
    MAED_CHECK_ARG(maed_x3p_shape_ok(a_hi, a_lo, lda, b_hi, b_lo, ldb, M, N, K), MAED_ERR_SHAPE,
                   "gemm_nt_planes: needs M, N > 0, K %% 32 == 0, leading dimensions %% 8 == 0, 16-byte aligned planes of less than 4 GB (M=%lld N=%lld K=%lld lda=%lld ldb=%lld)",
                   (long long)M, (long long)N, (long long)K, (long long)lda, (long long)ldb);
    MAED_CHECK_ARG(epilogue == MAED_EPI_STORE || epilogue == MAED_EPI_GELU || epilogue == MAED_EPI_RESID_F32, MAED_ERR_ARG,
                   "gemm_nt_planes: epilogue must be MAED_EPI_STORE, MAED_EPI_GELU or MAED_EPI_RESID_F32");
    MAED_CHECK_ARG(variant == 0 || (variant >= 2 && variant <= 7 && variant != 3), MAED_ERR_ARG, "gemm_nt_planes: variant must be 0 (default), 2, 4, 5, 6 or 7");
    MAED_CHECK_ARG(out || (out_hi && epilogue != MAED_EPI_RESID_F32), MAED_ERR_ARG, "gemm_nt_planes: no output (out may be NULL only when the planes are asked for)");
    MAED_CHECK_ARG(!out_lo || out_hi, MAED_ERR_ARG, "gemm_nt_planes: out_lo without out_hi");
    MAED_CHECK_ARG(ldo >= N, MAED_ERR_SHAPE, "gemm_nt_planes: ldo < N");
    if (epilogue == MAED_EPI_RESID_F32) MAED_CHECK_ARG(aux && ldaux >= N && !out_hi, MAED_ERR_ARG, "gemm_nt_planes: MAED_EPI_RESID_F32 needs aux (fp32) and writes fp32 only");
    // (the epilogues store / load 16 and 32 bytes per lane whenever ldo % 8 == 0 and ldaux % 8 == 0: a misaligned output from a C host must be an error, not a fault)
    MAED_CHECK_ARG(is_aligned(out, 16) && is_aligned(out_hi, 16) && is_aligned(out_lo, 16) && is_aligned(out2_bf16, 16) && is_aligned(aux, 16) && is_aligned(bias, 16),
                   MAED_ERR_ALIGN, "gemm_nt_planes: out, out_hi, out_lo, out2_bf16, aux and bias must be 16-byte aligned");
    EpiArgs e{};
    e.bias = bias; e.out = out; e.ldo = ldo; e.out2 = out2_bf16; e.aux = aux; e.ldaux = ldaux; e.twin = out_hi; e.lo = out_lo; e.out2_bf16 = true;
    MAED_PROPAGATE(maed_gemm_nt_x3p_launch(epilogue, variant, a_hi, a_lo, lda, b_hi, b_lo, ldb, M, N, K, e, (hipStream_t)stream));
    MAED_CHECK_LAUNCH("gemm_nt_planes");
    return MAED_OK;
}

extern "C" int maed_split_planes(const float* x, void* hi, void* lo, int64_t n, void* stream) {
    MAED_CHECK_ARG(n >= 0 && n % 8 == 0, MAED_ERR_SHAPE, "split_planes: n must be a multiple of 8");
    if (n == 0) return MAED_OK;
    MAED_CHECK_ARG(x && hi && lo, MAED_ERR_ARG, "split_planes: null pointer");
    MAED_CHECK_ARG(is_aligned(x, 16) && is_aligned(hi, 16) && is_aligned(lo, 16), MAED_ERR_ALIGN, "split_planes: pointers must be 16-byte aligned");
    const float* src[1] = {x}; void* h[1] = {hi}; void* l[1] = {lo}; const int64_t nn[1] = {n};
    MAED_PROPAGATE(maed_split_planes_launch(1, src, h, l, nn, (hipStream_t)stream));
    MAED_CHECK_LAUNCH("split_planes");
    return MAED_OK;
}
