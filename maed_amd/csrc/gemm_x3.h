// Internal interface of csrc/gemm_x3.hip (split-bf16 "bf16x3 / bf16x6" GEMMs on fp32 operands) towards the extern "C" entry points of
// gemm.hip / gemm_tn.hip / attention.  Not part of the C-ABI.
#pragma once
#include "common.cuh"
#include "gemm_epilogue.cuh"

// 3x3 implicit GEMM geometry: as gemm.hip's Conv3x3Dims.  B element (n, tap, c) = W[b_base + tap * b_tap + n * b_row + c]
struct X3ConvDims { int F, H, W, Cin, Ho, Wo, stride, pad_top, pad_left; int64_t b_row, b_tap, b_base; };
// weight gradient of the stride-1 3x3 convolution: per-pixel 9-bit "tap inside the image" mask (maed_conv3x3_tapmask), Cin, image width
struct X3TnConv { const uint16_t* tapmask; int Cin, Wimg; };

// number of bf16 planes of the process-wide fp32 matmul mode (maed_set_option(MAED_OPT_F32_MATMUL)): 0 = exact VALU kernels, 2 = bf16x3, 3 = bf16x6
int maed_x3_planes(void);

bool maed_x3_nt_shape_ok(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t K);
int maed_gemm_nt_x3_launch(int epilogue, int np, const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const EpiArgs& e,
                           int splitk, hipStream_t s);
int maed_conv1x1_x3_launch(int np, const void* x, int64_t ldx, const void* w, int64_t ldw, int64_t M, int Cout, int Cin, const EpiArgs& e, bool gn, hipStream_t s);
int maed_conv3x3_x3_launch(int np, const void* x, const void* w, const X3ConvDims& d, int64_t M, int Cout, const EpiArgs& e, bool add, bool gn, hipStream_t s);
int maed_gemm_tn_x3_launch(int np, const void* Y, int64_t ldy, const void* X, int64_t ldx, int64_t M, int N, int K, float* dW, int64_t ldw, float* dbias,
                           const X3TnConv* conv, int target_wgs, hipStream_t s);
