// Internal interface of csrc/gemm_x3.hip (split-bf16 "bf16x3 / bf16x6" GEMMs on fp32 operands) towards the extern "C" entry points of
// gemm.hip / gemm_tn.hip / attention.  Not part of the C-ABI.
#pragma once
#include "common.cuh"
#include "gemm_epilogue.cuh"

// 4 consecutive fp32 -> NP planes of 4 bf16 (2 dwords per plane): x = x0 + x1 (+ x2), x0 = bf16(x) (round to nearest even), x1 = bf16(x - x0), ...
// (every subtraction is exact in fp32)
// v_cvt_pk_bf16_f32 as an opaque operation: through the __bf16 vector cast hipcc re-converts the LOW element on its own to form float(x0) (it folds
// "(pack << 16)" back into a scalar conversion): 6 converter instructions per 4 values instead of 4 (ISA of the first build, profiles/r03_isa_audit_x3.txt)
__device__ __forceinline__ uint32_t x3_pack_bf2(float lo, float hi) {
#ifdef MAED_HOSTSIM
    return pack_bf2(lo, hi);
#else
    uint32_t w;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(lo), "v"(hi));
    return w;
#endif
}
typedef float x3_f32x2_t __attribute__((ext_vector_type(2)));
template <int NP>
__device__ __forceinline__ void split4(float r0, float r1, float r2, float r3, uint2 (&pl)[NP]) {
    x3_f32x2_t a = {r0, r1}, b = {r2, r3};                   // (pairs: the residual subtraction is one v_pk_add_f32 per pair)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const uint32_t w0 = x3_pack_bf2(a.x, a.y), w1 = x3_pack_bf2(b.x, b.y);
        pl[p] = make_uint2(w0, w1);
        if (p + 1 < NP) {
            const x3_f32x2_t ha = {__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u)}, hb = {__uint_as_float(w1 << 16), __uint_as_float(w1 & 0xffff0000u)};
            a -= ha;
            b -= hb;
        }
    }
}
// the MFMAs of one split product: sum over the plane pairs whose orders add up to less than NP, smallest partial products first
template <int NP>
__device__ __forceinline__ f32x16_t mfma_split(const bf16x8_t (&a)[NP], const bf16x8_t (&b)[NP], f32x16_t acc) {
#pragma unroll
    for (int ord = NP - 1; ord >= 0; --ord)
#pragma unroll
        for (int pa = 0; pa <= ord; ++pa) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa], b[ord - pa], acc, 0, 0, 0);
    return acc;
}

// 3x3 implicit GEMM geometry: as gemm.hip's Conv3x3Dims.  B element (n, tap, c) = W[b_base + tap * b_tap + n * b_row + c]
struct X3ConvDims { int F, H, W, Cin, Ho, Wo, stride, pad_top, pad_left; int64_t b_row, b_tap, b_base; };
// weight gradient of the stride-1 3x3 convolution: per-pixel 9-bit "tap inside the image" mask (maed_conv3x3_tapmask), Cin, image width
struct X3TnConv { const uint16_t* tapmask; int Cin, Wimg; };

// number of bf16 planes of the process-wide fp32 matmul mode (maed_set_option(MAED_OPT_F32_MATMUL)): 0 = exact VALU kernels, 2 = bf16x3, 3 = bf16x6
int maed_x3_planes(void);
int maed_tn_splits(int tiles);          // gemm_tn.hip: M-split heuristic of the weight-gradient GEMMs
// dtype codes MAED_F32X3 / MAED_F32X6 = fp32 storage with an explicit engine: returns the plane count they ask for (0 for plain MAED_F32 / MAED_BF16)
// and rewrites `dtype` to MAED_F32
static inline int maed_x3_take_dtype(int& dtype) {
    if (dtype == MAED_F32X3) { dtype = MAED_F32; return 2; }
    if (dtype == MAED_F32X6) { dtype = MAED_F32; return 3; }
    if (dtype == MAED_F32X1) { dtype = MAED_F32; return 1; }
    return 0;
}

bool maed_x3_nt_shape_ok(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t K);
int maed_gemm_nt_x3_launch(int epilogue, int np, const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const EpiArgs& e,
                           int splitk, hipStream_t s);
int maed_conv1x1_x3_launch(int np, const void* x, int64_t ldx, const void* w, int64_t ldw, int64_t M, int Cout, int Cin, const EpiArgs& e, bool gn, hipStream_t s);
int maed_conv3x3_x3_launch(int np, const void* x, const void* w, const X3ConvDims& d, int64_t M, int Cout, const EpiArgs& e, bool add, bool gn, hipStream_t s);
// csrc/attn_x3.hip: the K/V-tiled attention kernels of attn_long.hip on fp32 operands with split-bf16 contractions (np = 2 / 3 planes)
int maed_attn_x3_fwd_launch(int np, const void* qkv, void* o, float* lse, int F, int L, int H, float scale, hipStream_t s);
int maed_attn_x3_bwd_launch(int np, const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int accumulate, int F, int L, int H, float scale,
                            hipStream_t s);
// temporal attention on one-tile virtual sequences (32 % T == 0), two planes only: false = shape / engine not covered (the caller runs its exact kernels)
bool maed_attn_tm_x3_fwd_launch(int np, const void* qkv, void* o, float* lse, int F, int P, int H, int T, float scale, hipStream_t s);
bool maed_attn_tm_x3_bwd_launch(int np, const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int accumulate, int F, int P, int H, int T,
                                float scale, hipStream_t s);
int maed_gemm_tn_x3_launch(int np, const void* Y, int64_t ldy, const void* X, int64_t ldx, int64_t M, int N, int K, float* dW, int64_t ldw, float* dbias,
                           const X3TnConv* conv, int target_wgs, hipStream_t s);

// csrc/gemm_x3p.hip: the two-plane NT product on operands stored as (hi, lo) bf16 planes; variant 0 = MAED_OPT_X3_PLANES's ring, 2 / 4 = stages
bool maed_x3p_shape_ok(const void* Ah, const void* Al, int64_t lda, const void* Bh, const void* Bl, int64_t ldb, int64_t M, int64_t N, int64_t K);
int maed_gemm_nt_x3p_launch(int epilogue, int variant, const void* a_hi, const void* a_lo, int64_t lda, const void* b_hi, const void* b_lo, int64_t ldb, int64_t M,
                            int64_t N, int64_t K, const EpiArgs& e, hipStream_t s);
int maed_split_planes_launch(int count, const float* const* src, void* const* hi, void* const* lo, const int64_t* n, hipStream_t s);   // count <= 8
