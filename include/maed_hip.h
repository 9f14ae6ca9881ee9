/*
 * maed_hip.h -- C-ABI of libmaed_hip.so: the MI355X (gfx950) kernels behind the MAED hot path.
 *
 * The reference (ziniuwan/maed) has no FFI layer: its hot path is a composition of ATen ops inside
 * lib/models (Python).  Each entry point below replaces one such composition; the comment on every
 * function cites the reference lines (relative to the reference repo root) it stands in for.
 * The host side (the maed_amd Python package) binds these with ctypes from torch.autograd.Functions that sit
 * inside nn.Modules carrying the reference's class / attribute / state_dict names.
 *
 * Conventions
 *  - plain pointers and extents only; no torch types.  All pointers are DEVICE pointers unless the
 *    name ends in _host.  The caller owns every buffer (inputs, outputs, workspaces); nothing here
 *    allocates, frees or synchronises.  Every launch goes to the hipStream_t passed as `stream`.
 *  - return value: MAED_OK (0) or a negative maed_status.  Never throws, never aborts.
 *    maed_last_error() returns a thread-local human readable message for the last failure.
 *  - `dtype` selects the storage/compute type of activations and GEMM weights:
 *    MAED_F32 (exact-f32 VALU kernels: the parity mode) or MAED_BF16 (MFMA kernels, fp32
 *    accumulation: the throughput mode).  LayerNorm parameters, biases, the residual stream,
 *    softmax statistics and all parameter gradients are always fp32.
 *  - row-major everywhere; "ld" arguments are leading dimensions in ELEMENTS.
 *  - token layout: tokens[(f*P + p)*C + c], f = n*T + t (frames of a clip are contiguous),
 *    qkv[(f*P + p)*3C + s*C + h*64 + e]   (vision_transformer.py:147 channel order s, h, e).
 *  - head dimension is 64 in every supported configuration (C = 64*H).
 */
#ifndef MAED_HIP_H
#define MAED_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { MAED_F32 = 0, MAED_BF16 = 1,
               /* fp32 STORAGE with an explicit matrix-product engine (the per-call form of MAED_OPT_F32_MATMUL; accepted by the matrix-product entry
                * points maed_gemm_nt, maed_gemm_tn_wgrad, maed_conv1x1_fwd, maed_conv3x3_fwd, maed_conv3x3_wgrad): split-bf16 with 3 / 6 MFMAs per product */
               MAED_F32X3 = 2, MAED_F32X6 = 3,
               /* fp32 storage, ONE bf16 plane per operand (one MFMA per product: bf16-level products, fp32 accumulation, no bf16 copy of the tensors): the
                * BACKWARD products of the mixed mode "bf16x3 forward / bf16 backward" (maed_gemm_nt, maed_gemm_tn_wgrad, maed_conv3x3_fwd as an input gradient,
                * maed_conv3x3_wgrad; the attention entry points keep two planes) */
               MAED_F32X1 = 4 } maed_dtype;

typedef enum {
    MAED_OK = 0,
    MAED_ERR_ARG = -1,         /* null pointer / bad enum */
    MAED_ERR_SHAPE = -2,       /* unsupported extent */
    MAED_ERR_ALIGN = -3,       /* pointer or leading dimension not aligned for vector access */
    MAED_ERR_LAUNCH = -4,      /* hipLaunch failure (message holds hipGetErrorString) */
    MAED_ERR_UNSUPPORTED = -5  /* combination not implemented */
} maed_status;

/* GEMM epilogues (out = epilogue(acc = A * B^T)) */
typedef enum {
    MAED_EPI_STORE = 0,      /* out[T]   = acc + bias                                              */
    MAED_EPI_GELU = 1,       /* out2[T]  = acc + bias ; out[T] = gelu_erf(out2)  (nn.GELU); out2 = NULL: only out is written */
    MAED_EPI_RESID_F32 = 2,  /* out[f32] = aux[f32] + acc + bias   (residual add fused)             */
    MAED_EPI_MUL_DGELU = 3,  /* out[T]   = acc * gelu_erf'(aux[T]) (backward through GELU)          */
    MAED_EPI_ATOMIC_F32 = 4, /* out[f32] += acc  (atomic; weight gradients, split-K allowed)        */
    MAED_EPI_STORE_F32 = 5,  /* out[f32] = acc + bias                                               */
    MAED_EPI_TANH = 6,       /* out[T]   = tanh(acc + bias)                                         */
    MAED_EPI_ADD = 7         /* out[T]   = aux[T] + acc + bias  (gradient accumulation at a residual fork); out2 != NULL: aux is first
                              * masked by 1 bit per element (bit c & 7 of byte (r * ldaux + c) / 8 = maed_groupnorm_fwd's relu_mask: the
                              * masked residual gradient of a GroupNorm + residual + ReLU is then never materialised) */
} maed_epilogue;

/* kernel implementation selector for ops that have both */
typedef enum { MAED_IMPL_AUTO = 0, MAED_IMPL_VALU = 1, MAED_IMPL_MFMA = 2,
               MAED_IMPL_MFMA_GLDS1 = 3, MAED_IMPL_MFMA_GLDS2 = 4, /* gemm_nt only: direct global->LDS staging, 1 or 2 LDS buffers */
               MAED_IMPL_MFMA_LONG = 5, /* attn_spatial only: K/V-tiled long-sequence kernels (chosen automatically past 512 / 320 tokens) */
               MAED_IMPL_MFMA_256 = 6, /* gemm_nt only: 256x256 tiles, counted-vmcnt LDS-DMA pipeline (csrc/gemm256.hip) */
               MAED_IMPL_X1 = 9, /* gemm_nt only: fp32 operands, ONE bf16 plane (what dtype MAED_F32X1 selects) */
               MAED_IMPL_MFMA_SK = 10, /* gemm_nt only: the persistent K-stream kernel (csrc/gemm_sk.hip), K cuts per MAED_OPT_SK (1 and 3: allowed) */
               MAED_IMPL_X3 = 7, MAED_IMPL_X6 = 8 /* MAED_F32 matrix products on the bf16 matrix cores: every fp32 operand split into 2 / 3 bf16
                                                   * terms, 3 / 6 MFMAs per product, fp32 accumulation (csrc/gemm_x3.hip): |error| ~2^-16 / ~2^-23
                                                   * of |a||b| instead of bf16's 2^-8.  MAED_IMPL_AUTO takes them when MAED_OPT_F32_MATMUL says so. */
} maed_impl;

const char* maed_last_error(void);
int maed_version(void);

/* ---- process-wide options -------------------------------------------------------------------------
 * The library reads no environment variables: the host sets what it wants after loading it.  Options are plain integers, may be changed
 * between calls (a call reads them once when it enqueues its kernels) and are safe to set from any host thread. */
typedef enum {
    MAED_OPT_F32_MATMUL = 0,    /* arithmetic of the matrix products (GEMMs, convolutions, attention contractions) of MAED_F32 calls with MAED_IMPL_AUTO:
                                 * 0 (default) exact fp32 FMA chains on the VALU -- the bit-for-bit parity mode;
                                 * 1 "bf16x3": operands split into two bf16 terms, three MFMAs per product, fp32 accumulation (error ~2^-16);
                                 * 2 "bf16x6": three terms, six MFMAs (error ~2^-23, fp32 level).
                                 * The analogue of torch.set_float32_matmul_precision: storage stays fp32, only the contraction engine changes.
                                 * Entry points that have no exact fp32 kernel (maed_conv1x1_fwd, maed_conv3x3_fwd, maed_gemm_tn_wgrad,
                                 * maed_conv3x3_wgrad) accept MAED_F32 only with 1 or 2. */
    MAED_OPT_SIDE_STREAM = 1,   /* 1 (default): weight-gradient GEMMs / temporal attention of the fused STE block on the library's side stream */
    MAED_OPT_TN_TARGET_WGS = 2, /* workgroups a weight-gradient GEMM is split into along M: 0 (default) = built-in heuristic, >= 64 = that target (sweep knob) */
    MAED_OPT_ABLATE = 3,        /* diagnostic builds (-DMAED_GEMM_ABLATE) only: bit mask of pipeline stages to drop */
    MAED_OPT_GN_BWD_ONEPASS = 4,/* 1 (default): maed_groupnorm_bwd with frame_sync reads x and dy once (register-resident slices + per-frame barrier); 0: two passes */
    MAED_OPT_F32_BWD_X1 = 5,    /* 1: the fused STE block's BACKWARD matrix products on fp32 tensors use one bf16 plane (MAED_F32X1) whatever the forward engine is;
                                 * 0 (default): the process-wide engine.  The host sets it together with its own per-call dtype codes (ops.set_float32_backward_precision) */
    MAED_OPT_ST_FUSED = 6,      /* 1 (default): the fused STE block runs the attentive addition as ONE launch per direction (maed_st_fused_fwd/bwd) where supported */
    MAED_OPT_CONV3X3_ROWS_WGS = 7, /* row-item 3x3 weight gradient at 64 -> 64 channels (maed_conv3x3_wgrad_rows64): workgroups per launch (default 256); 0: the shape
                                   * goes to the general TN kernel instead (A/B knob) */
    MAED_OPT_STEM_WGRAD_WGS = 8,   /* workgroups of maed_stem7x7s2_wgrad (default 512) */
    MAED_OPT_LBS_FRAMES = 9,       /* SMPL linear blend skinning: 0 (default) = the pose-corrective blend on the fp32 matrix cores + a streaming skinning pass (smpl.hip,
                                    * round 5); 4 / 8 / 16 = the VALU kernel with that many frames per workgroup (A/B knob; what the host simulator runs) */
    MAED_OPT_TN_DMA = 10,          /* 1 (default): bf16 weight-gradient GEMMs (maed_gemm_tn_wgrad, M % 32 == 0) on the LDS-DMA + transposing-read kernel (csrc/gemm_tn2.hip);
                                    * 0: the register-transposing kernel (csrc/gemm_tn.hip) -- A/B knob */
    MAED_OPT_X3_PLANES = 11,       /* maed_ste_block_fwd_twin: fc1's activation is stored as (hi, lo) bf16 planes -- hi is the backward's twin, 4 bytes per element instead of
                                    * fp32 + twin = 6 -- and fc2 runs on maed_gemm_nt_planes' kernel variant of this value: 6 (default) = 256 x 256 tiles, 2 / 4 = 128 x 128 with a
                                    * ring of 2 / 4 stages, 5 = 256 x 128, 7 = 128 x 128 with K tiles of 64; 0 = fp32 activation + twin, fc2 on the fp32-operand kernel (A/B knob) */
    MAED_OPT_X3_PLANES_LN = 12,    /* with MAED_OPT_X3_PLANES != 0: 1 = the block's two LayerNorm outputs are stored as planes too (hi straight into the bf16 arena: no cast pass
                                    * for them) and qkv / fc1 run on the 128 x 128 plane kernel; 0 (default) = fp32 LayerNorm outputs, fp32-operand kernel.  Measured neutral
                                    * in the cfg3 train step (28.83 vs 28.80 ms): kept as an A/B knob */
    MAED_OPT_SK = 13,              /* bf16 maed_gemm_nt on the persistent K-stream kernel (csrc/gemm_sk.hip: one workgroup per CU walks 256 x 256 tiles as one continuous
                                    * stream of K tiles; tiles that do not fill a round of the grid are cut along K, "stream-K") where the shape allows (K % 128 == 0,
                                    * M, N >= 256) and the dispatcher's heuristic picks it: 1 (default) = heuristic, 0 = never (the per-tile kernels of rounds 1-5),
                                    * 2 = wherever the shape allows without K cuts, 3 = wherever the shape allows, K cuts whenever the tiles do not fill whole rounds */
    MAED_OPT_SK_GRID = 14,         /* workgroups of that kernel: 0 (default) = one per CU; a smaller grid is a test / sweep knob */
    MAED_OPT_TN_SK = 15,           /* bf16 maed_gemm_tn_wgrad on the persistent K-stream kernel (csrc/gemm_tn_sk.hip: 256 x 256 tiles, the reduction rows dealt to one workgroup
                                    * per CU, partial tiles in slabs, a second launch adds them in a fixed order -- no atomics) where the shape allows (M % 128 == 0; N, K >= 256
                                    * and multiples of 128; at most one tile per two CUs): 1 (default) = yes, 0 = the split-M kernels with closing atomics (gemm_tn2.hip / gemm_tn.hip) */
    MAED_OPT_CONV3X3_NARROW_WGS = 16, /* maed_conv3x3_fwd (bf16): when the 128 x 128 tiling of the output gives fewer workgroups than this, the 128 x 64 tile is used
                                    * instead (twice the workgroups: the stage-3 convolutions of the R50 run at 1.5 workgroups per CU on 128 x 128 tiles and are
                                    * latency-bound for it).  0 = 128 x 64 only for Cout <= 64 (rounds 2-5). */
    MAED_OPT_CONV3X3_FRAME = 17,   /* maed_conv3x3_fwd (bf16, round 6): feature maps of 129 .. 256 pixels whose frames fill the chip (stage 3 of the R50: 14 x 14, 128 frames x
                                    * 2 column tiles) run on ONE FRAME x 128 channels per workgroup with a three-stage copy ring (one workgroup per CU, no imbalance, the
                                    * weight rows shared by a whole frame): 1 (default); 0 = the 128 x 128 tiles of rounds 2-5 (A/B knob); 2 = whenever the shape allows, however few frames (tests) */
    MAED_OPT_COUNT
} maed_option;
/* Check that `device` (a HIP device ordinal) is one this library was built for (gfx950: MI355X).  MAED_OK, or MAED_ERR_UNSUPPORTED with the device's
 * architecture in maed_last_error() -- the host calls it once after loading the library, so that a wrong device fails here and not as an "invalid device
 * function" at the first launch.  It also creates, here and nowhere else, every runtime object the library owns: two non-blocking side streams, two rings of
 * timing-less events (fences between the caller's stream and the side streams) and one word of pinned host memory (maed_device_faults); nothing is created at call
 * time afterwards (a host that never called maed_init gets the same objects on first use; the opt-in communicator has its own maed_comm_init).  One device per
 * process (one process per GPU, as train.py:166-182 launches them). */
int maed_init(int device);
/* Frame-barrier timeouts since the process started (or since maed_device_faults_clear).  The one-pass GroupNorm backward (MAED_OPT_GN_BWD_ONEPASS) and the fused
 * attentive addition (MAED_OPT_ST_FUSED) synchronise the workgroups of one frame INSIDE a launch; that needs them co-resident, which holds on a GPU this process has
 * to itself.  When a peer does not arrive within the spin bound (GPU shared with another process, preemption, a debugger), the affected call's result is poisoned with
 * NaN (loud, never subtly wrong), the kernel raises this counter (pinned host memory, system-scope atomic: no synchronisation needed to read it), and from the next
 * call on both operations run as their multi-launch forms for the rest of the process; maed_last_error() carries the explanation.  Hosts that share GPUs should set
 * MAED_OPT_GN_BWD_ONEPASS = 0 and MAED_OPT_ST_FUSED = 0 up front (environment: MAED_GN_BWD_ONEPASS=0 MAED_ST_FUSED=0).  No reference counterpart (scheduling only). */
int maed_device_faults(void);
int maed_device_faults_clear(void);
int maed_set_option(int key, int value);   /* MAED_OK or MAED_ERR_ARG */
int maed_get_option(int key);              /* the value, or MAED_ERR_ARG (negative) for an unknown key */

/* ---- K1: nn.LayerNorm(eps=1e-6)  (vision_transformer.py:249,254,344,569) ---------------------- */
/* y[T](rows,C) = LN(x[f32]) ; saves mean/rstd (fp32, rows) for backward.  x rows may be strided. */
int maed_layernorm_fwd(const float* x, int64_t x_row_stride, const float* gamma, const float* beta,
                       void* y, int dtype, float* mean, float* rstd,
                       int64_t rows, int C, float eps, void* stream);
/* dx_out[f32] = (dres_in ? dres_in : 0) + LN'(dy) ; dgamma/dbeta += (fp32, atomics).
 * dx_twin (optional): the same rows in the compute dtype (the operand of the next backward GEMMs). */
int maed_layernorm_bwd(const void* dy, int dtype, const float* x, int64_t x_row_stride,
                       const float* gamma, const float* mean, const float* rstd,
                       const float* dres_in, float* dx_out, void* dx_twin, float* dgamma, float* dbeta,
                       int64_t rows, int C, void* stream);

/* ---- K2/K6/K7/K9/K10: nn.Linear as out = epi(A[M,K] * B[N,K]^T)  (vision_transformer.py:98-111,
 *      :124-128,:147,:154,:176; ktd.py:74-79)  B is the nn.Linear weight as stored (out,in). -------- */
int maed_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb,
                 int64_t M, int64_t N, int64_t K, int dtype, int epilogue,
                 const float* bias, void* out, int64_t ldo, void* out2,
                 const void* aux, int64_t ldaux, int splitk, int impl, void* stream);

/* The same product in the accurate mode's arithmetic ("bf16x3": 16 significand bits per operand, three bf16 MFMAs per product, fp32 accumulators) on operands that are
 * STORED as two bf16 planes, x = hi + lo with hi = bf16(x) (round to nearest even), lo = bf16(x - hi) (round 5; csrc/gemm_x3p.hip):
 *     out[f32](M, ldo) = epi((a_hi + a_lo)[M,K] (b_hi + b_lo)[N,K]^T)          both planes of an operand share its leading dimension (elements)
 * bit for bit what maed_gemm_nt computes with dtype MAED_F32X3 on the fp32 operand the planes were split from; the planes move the same 4 bytes per element but reach
 * LDS by LDS-DMA (no register staging, no VALU split), and the hi plane doubles as the bf16 twin a bf16 backward reads (maed_ste_block_fwd_twin).
 * epilogue: MAED_EPI_STORE (+ bias), MAED_EPI_GELU (+ bias; out2_bf16 = the pre-activation as bf16, or NULL), MAED_EPI_RESID_F32 (out = aux[f32] + product + bias).
 * out_hi / out_lo (STORE, GELU; leading dimension ldo): the result as planes for the next product of the chain; `out` may be NULL when out_hi is given.
 * variant: 0 = the one MAED_OPT_X3_PLANES names; 2 / 4 = 128 x 128 tiles with a copy ring of 2 / 4 stages, 5 = 256 x 128 (3 stages), 6 = 256 x 256 (2 stages),
 * 7 = 128 x 128 with K tiles of 64 on eight waves -- all bit-identical (profiles/r05_x3p_micro.txt).  K % 32 == 0, lda / ldb % 8 == 0, planes 16-byte aligned and < 4 GB each.
 * Reference: the nn.Linear calls of vision_transformer.py:98-111,124-128 in fp32. */
int maed_gemm_nt_planes(const void* a_hi, const void* a_lo, int64_t lda, const void* b_hi, const void* b_lo, int64_t ldb, int64_t M, int64_t N, int64_t K,
                        int epilogue, const float* bias, void* out, int64_t ldo, void* out2_bf16, const void* aux, int64_t ldaux, void* out_hi, void* out_lo,
                        int variant, void* stream);
/* hi[bf16](n), lo[bf16](n) = the two planes of x[f32](n) (n % 8 == 0, 16-byte aligned): what maed_gemm_nt_planes takes for a tensor that exists as fp32 (weights) */
int maed_split_planes(const float* x, void* hi, void* lo, int64_t n, void* stream);

/* weight gradient dW[N,K] += Y[M,N]^T X[M,K] and (optional) bias gradient dbias[N] += colsum(Y), bf16 operands, fp32 atomics (split over M).  No transposed
 * copies in memory.  Default for M % 32 == 0 (MAED_OPT_TN_DMA = 1): csrc/gemm_tn2.hip -- both operands copied unchanged into LDS by LDS-DMA, fragments by
 * ds_read_b64_tr_b16, the bias gradient as one more MFMA against a ones operand; otherwise, or with MAED_OPT_TN_DMA = 0: csrc/gemm_tn.hip -- 8x8 blocks transposed in
 * registers while staging.  Replaces autograd's backward of nn.Linear / the 1x1 convolutions (vision_transformer.py:98-111,124-128; resnetv2.py:91-93). */
int maed_gemm_tn_wgrad(const void* Y, int64_t ldy, const void* X, int64_t ldx, int64_t M, int N, int K,
                       float* dW, int64_t ldw, float* dbias, int dtype, void* stream);

/* transpose helper feeding the weight-gradient GEMMs of the f32 parity mode (dW = dY^T X as an NT GEMM on transposed
 * copies): out_t[T](N, ldt) = in(M,N)^T with columns [M, ldt) zero-filled; optional plain cast copy
 * out_c[T](M,N); optional colsum[N] += sum over rows (bias gradient).  in_dtype may be MAED_F32
 * while dtype (of the outputs) is MAED_BF16.  */
int maed_transpose_cast(const void* in, int in_dtype, int64_t ldi, int64_t M, int64_t N,
                        void* out_t, int64_t ldt, void* out_c, int64_t ldc, float* colsum,
                        int dtype, void* stream);

/* ---- K3: Attention.forward_spatial (vision_transformer.py:206-214) --------------------------- */
/* per (frame, head): o = softmax(q k^T * scale) v over the P tokens of a frame.
 * qkv (F,P,3C) as written by the qkv Linear; o (F,P,C) head-major channels; lse (F,H,P) fp32 =
 * log-sum-exp of the scaled scores (saved for backward). */
int maed_attn_spatial_fwd(const void* qkv, void* o, float* lse, int F, int P, int H, float scale,
                          int dtype, int impl, void* stream);
/* dqkv (F,P,3C): q/k/v gradients; accumulate!=0 adds into dqkv instead of overwriting. */
int maed_attn_spatial_bwd(const void* qkv, const void* o, const void* d_o, const float* lse,
                          void* dqkv, int accumulate, int F, int P, int H, float scale,
                          int dtype, int impl, void* stream);

/* ---- K4: Attention.forward_temporal (vision_transformer.py:216-228) -------------------------- */
/* per (clip, head, token): attention across the T frames of the clip; no transposed copies.
 * dtype MAED_F32 in the bf16x3 mode (MAED_OPT_F32_MATMUL = 1, or dtype = MAED_F32X3 for this call) with 32 % T == 0: split-bf16 contractions on the
 * matrix cores (csrc/attn_x3.hip); every other fp32 case (other T, bf16x6) runs the exact fp32 kernels. */
int maed_attn_temporal_fwd(const void* qkv, void* o, float* lse, int F, int P, int H, int T,
                           float scale, int dtype, void* stream);
int maed_attn_temporal_bwd(const void* qkv, const void* o, const void* d_o, const float* lse,
                           void* dqkv, int accumulate, int F, int P, int H, int T, float scale,
                           int dtype, void* stream);

/* ---- K5: attentive addition (vision_transformer.py:152-158) ----------------------------------- */
/* means[T](F,2C) = mean over tokens of [x_s || x_t];  ws: caller-owned fp32 scratch of F*2C floats */
int maed_st_colmean(const void* x_s, const void* x_t, void* means, float* ws, int F, int P, int C, int dtype, void* stream);
/* mix[T](F,P,C) = x_t*a1 + x_s*a0, (a0,a1) = softmax(logits[f][2c], logits[f][2c+1]); logits fp32 (F,2C) */
int maed_st_mix_fwd(const void* x_s, const void* x_t, const float* logits, void* mix,
                    int F, int P, int C, int dtype, void* stream);
/* dlogits[T](F,2C) from dmix: reduction over tokens + 2-way softmax backward; ws: F*2C fp32 scratch */
int maed_st_mix_bwd_reduce(const void* dmix, const void* x_s, const void* x_t, const float* logits,
                           void* dlogits, float* ws, int F, int P, int C, int dtype, void* stream);
/* dx_s = dmix*a0 + dmeans[f][c]/P ; dx_t = dmix*a1 + dmeans[f][C+c]/P ; dmeans[T](F,2C) */
int maed_st_mix_bwd_apply(const void* dmix, const float* logits, const void* dmeans, void* dx_s, void* dx_t,
                          int F, int P, int C, int dtype, void* stream);
/* K5 in ONE launch per direction (bf16; vision_transformer.py:152-158,176): token means + ts_attn Linear + pair softmax + mix fused -- a workgroup owns 128 channels
 * of one frame for all P tokens in registers, the C / 128 workgroups of a frame exchange their means (forward) / dlogits (backward) through `ex` and meet at
 * `sync`.  Writes what the separate entry points write: means (F,2C) bf16, logits (F,2C) fp32, mix; backward: dlogits (F,2C) bf16 (operand of the ts_attn weight
 * gradient, which stays a maed_gemm_tn_wgrad call), dx_s, dx_t.  w_ts (2C,2C) as nn.Linear stores it, wt_ts its transposed image.
 * sync: F*16 uint32 (cleared by the call), ex: F*2C floats -- caller-owned scratch.  maed_st_fused_supported: bf16, C % 128 == 0, C <= 1024, P <= 288. */
int maed_st_fused_supported(int P, int C, int dtype);
int maed_st_fused_fwd(const void* x_s, const void* x_t, const void* w_ts, const float* b_ts, void* means, float* logits, void* mix,
                      uint32_t* sync, float* ex, int F, int P, int C, int dtype, void* stream);
int maed_st_fused_bwd(const void* dmix, const void* x_s, const void* x_t, const float* logits, const void* wt_ts, void* dlogits, void* dx_s,
                      void* dx_t, uint32_t* sync, float* ex, int F, int P, int C, int dtype, void* stream);


/* ---- K8: cls / pos / temporal embeddings (vision_transformer.py:392-399) ---------------------- */
/* tokens[f32](F,P,C): row 0 = cls, rows 1.. = patch[T](F,P-1,C); + pos_embed[p] + temp_embed[f % T] */
int maed_embed_add_fwd(const void* patch, int dtype, const float* cls, const float* pos, const float* temp,
                       float* tokens, int F, int P, int C, int T, void* stream);
/* dpatch[T](F,P-1,C) = dtokens rows 1.. ; dpos[f32](P,C) += sum_f dtokens[f] (row 0 of it is the cls token's gradient) ; dtemp[f32](T,C) += sum over the
 * tokens of every frame f with f % T = t  (both accumulators zeroed by the caller: the reductions are split over workgroups and meet with atomics) */
int maed_embed_add_bwd(const float* dtokens, void* dpatch, int dtype, float* dpos, float* dtemp, int F, int P, int C, int T, void* stream);

/* ---- element-wise pieces of the training tail -------------------------------------------------- */
/* nn.Dropout(p) in training (ktd.py:54,56): y[f32] = keep ? x / (1 - p) : 0 with keep = hash(seed, index) >= p; the same call with the
 * same seed applied to dy is the backward (nothing is stored).  x == y (in place) is allowed. */
int maed_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream);
/* Per-step scalars of a training step kept in DEVICE memory (round 6: hipGraph capture of the whole step, maed_amd/graphed.py).  A captured launch replays with
 * the arguments it was recorded with: whatever changes from step to step -- the learning rate the host's scheduler chose (train.py:123-127), Adam's bias
 * corrections 1 - beta^t, the seed of the Dropout masks (ktd.py:54,56) -- is read from this caller-owned 32-byte record instead, which the host rewrites
 * (one small copy on the same stream) before every replay.  The eager step takes the same entry points when it is handed a record, so both forms run the
 * same arithmetic. */
typedef struct { float lr, bias_corr1, bias_corr2, reserved0; uint64_t seed; uint64_t reserved1; } maed_train_state;
/* maed_dropout with seed = state->seed + call_id * 0x9E3779B97F4A7C15 (call_id: which Dropout layer of the step; its backward passes the same value) */
int maed_dropout_dev(const float* x, float* y, int64_t n, float p, const maed_train_state* state, uint64_t call_id, void* stream);
/* backward of the GEMM's TANH epilogue (pre_logits, vision_transformer.py:350-353): dx[T] = dy[f32] * (1 - y[T]^2) */
int maed_tanh_bwd(const float* dy, const void* y, void* dx, int64_t n, int dtype, void* stream);

/* ---- whole STE Block (vision_transformer.py:244-261 with Attention 'parallel' :146-158,176 and
 *      Mlp :106-112) as ONE host call that enqueues every kernel of the block -------------------- */
typedef struct {
    int F, P, C, H, T, hidden; /* frames, tokens/frame, embed dim, heads, clip length, MLP hidden */
    int dtype;                 /* maed_dtype of activations + GEMM weights */
    int impl;                  /* maed_impl for attention / GEMM */
    float eps;                 /* LayerNorm eps (1e-6) */
} maed_block_dims;

typedef struct {              /* fp32 unless noted; w_* are [out,in] in compute dtype */
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    const void *w_qkv, *w_ts, *w_proj, *w_fc1, *w_fc2;
    const float *b_qkv, *b_ts, *b_proj, *b_fc1, *b_fc2;
    /* transposed weights [in,out] in compute dtype (backward only; may be NULL for forward) */
    const void *wt_qkv, *wt_ts, *wt_proj, *wt_fc1, *wt_fc2;
} maed_block_params;

typedef struct {              /* fp32 gradient accumulators (+=) with the parameters' shapes */
    float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    float *w_qkv, *w_ts, *w_proj, *w_fc1, *w_fc2;
    float *b_qkv, *b_ts, *b_proj, *b_fc1, *b_fc2;
} maed_block_grads;

/* bytes of activations the forward saves for the backward / of transient backward scratch */
size_t maed_ste_block_saved_bytes(const maed_block_dims* d);
size_t maed_ste_block_scratch_bytes(const maed_block_dims* d);
/* x_out[f32](F,P,C) = Block(x_in[f32]).  `saved` (maed_ste_block_saved_bytes) is written and must be
 * handed unchanged to maed_ste_block_bwd.  x_out may alias x_in only if no backward is wanted. */
int maed_ste_block_fwd(const maed_block_dims* d, const maed_block_params* p, const float* x_in,
                       float* x_out, void* saved, void* stream);
/* "bf16x3 forward / bf16 backward from bf16 twins" (round 5).  d->dtype = MAED_F32 describes the FORWARD: fp32 activations, products on the process-wide fp32
 * engine (MAED_OPT_F32_MATMUL; p->w_* = the fp32 master weights, wt_* unused), written to the transient `work_f32` (maed_ste_block_twin_work_bytes).  What the
 * backward reads is saved to `saved_bf16` in the layout of a MAED_BF16 block (maed_ste_block_saved_bytes of the same dims with dtype MAED_BF16): bf16 twins of the
 * activations, the fp32 statistics as they are.  The backward is then maed_ste_block_bwd with dtype MAED_BF16 and bf16 weight images on that arena -- the throughput
 * mode's backward behind the accurate mode's forward.  The cast pass that writes the twins runs on a side stream beside the NEXT block's forward: consecutive calls
 * alternate work_slot 0 / 1 and give each slot its own work buffer (a slot's buffer is reused only after the cast that read it; maed_ste_block_bwd waits for the
 * outstanding casts).  Same reference lines as maed_ste_block_fwd. */
size_t maed_ste_block_twin_work_bytes(const maed_block_dims* d);
int maed_ste_block_fwd_twin(const maed_block_dims* d, const maed_block_params* p, const float* x_in, float* x_out, void* saved_bf16, void* work_f32, int work_slot,
                            void* stream);
/* `stream` waits for the cast passes of earlier maed_ste_block_fwd_twin calls that are still pending on the library's side stream.  maed_ste_block_bwd does this by
 * itself; a host calls it when NO backward will follow a twin forward (the arenas are about to be freed) and before it frees or re-sizes a work buffer.  No reference
 * counterpart (scheduling only). */
int maed_ste_block_twin_join(void* stream);
/* the same forward when no backward will follow (inference): `work` = maed_ste_block_saved_bytes of scratch; what only the backward would read
 * (fc1's pre-activation: 103 MB per block at cfg3) is not written.  x_out may alias x_in. */
int maed_ste_block_infer(const maed_block_dims* d, const maed_block_params* p, const float* x_in,
                         float* x_out, void* work, void* stream);
/* dx_in[f32] = dBlock/dx_in(dx_out); parameter gradients are accumulated into g. dx_in may alias dx_out.
 * dx_out_twin (optional, in): dx_out already in the compute dtype; dx_in_twin (optional, out): dx_in in the compute
 * dtype -- consecutive blocks hand the bf16 copy along so no cast pass is needed. */
int maed_ste_block_bwd(const maed_block_dims* d, const maed_block_params* p, const maed_block_grads* g,
                       const float* x_in, const float* dx_out, float* dx_in, void* saved, void* scratch,
                       const void* dx_out_twin, void* dx_in_twin, void* stream);

/* stream ordering helper for hosts that put single launches on a second stream of their own: everything enqueued on from_stream so far happens before whatever
 * is enqueued on to_stream from now on (one event record + one stream wait; the events come from a ring the library owns). */
int maed_stream_fence(void* from_stream, void* to_stream);

/* measurement only: bracket selected launches inside the composite block calls with hipEvents on the
 * launch stream.  Tags: 0 attn_spatial_fwd, 1 attn_temporal_fwd, 2 qkv GEMM, 3 fc1 GEMM, 4 fc2 GEMM,
 * 5 attn_spatial_bwd, 6 attn_temporal_bwd, 7 weight-gradient GEMMs, 8 proj GEMM, 9 the four input-gradient GEMMs of a block,
 * 10 (reserved: LayerNorm).  collect() synchronises the events and fills ms_total[n] / count[n] (host pointers,
 * n = maed_prof_ntags()), then clears the records. */
int maed_prof_enable(int on);
int maed_prof_ntags(void);
int maed_prof_collect(double* ms_total_host, int* count_host);
/* tags 11 / 12: EVERY maed_gemm_tn_wgrad / maed_conv3x3_wgrad launch (STE and backbone).  maed_prof_flops fills flops_host[n] with the algorithmic FLOPs
 * (2 M N K per launch) the tagged launches declared since the last collect -- call it before maed_prof_collect, which clears them. */
int maed_prof_flops(double* flops_host);
/* per-launch records of one tag in launch order (before maed_prof_collect): duration [us], declared FLOPs, algorithmic HBM bytes (operands read once, result written
 * once); returns the number of records held, writes at most cap of them.  bench.py prices every launch against ITS bound (MFMA or HBM) with these. */
int maed_prof_records(int tag, double* us_host, double* flops_host, double* bytes_host, int cap);

/* ---- K10: KTD joint chain (ktd.py:81-86) ------------------------------------------------------- */
/* base[f32](F,144) = x W_feat^T + b for the 1024-wide feature part of all 24 regressors has been
 * computed by maed_gemm_nt; this adds the ancestor terms serially along ANCESTOR_INDEX (ktd.py:10-35):
 * pose[f][6j..6j+5] = base[f][6j..] + sum_a W_anc[j][:, 6*slot(a)..] pose[f][6a..].
 * w_anc: packed fp32, for joint j a (6, 6*n_anc(j)) row-major block at anc_offset[j]. */
int maed_ktd_chain_fwd(const float* base, const float* w_anc, float* pose, int F, void* stream);

/* ---- K11: rot6d_to_rotmat + rotation_matrix_to_angle_axis (geometry.py:320-334,58-87,143-223,90-140) */
int maed_rot6d_pose_fwd(const float* pose6d, float* rotmat, float* angle_axis, int64_t n_joints, void* stream);

/* ---- K12-K15: SMPL (smplx 0.1.13 lbs, pose2rot=False; smpl.py:94-106; ktd.py:108-114; spin.py:113-157) */
typedef struct {
    const float* v_template;  /* (6890,3)      */
    const float* shapedirs;   /* (6890,3,10)   */
    const float* posedirs;    /* (207,20670)   */
    const float* J_template;  /* (24,3)  = J_regressor v_template   (precomputed by the host)  */
    const float* J_shapedirs; /* (24,3,10) = J_regressor shapedirs  (precomputed by the host)  */
    const float* lbs_weights; /* (6890,24)     */
    const int32_t* parents;   /* (24) root -1  */
} maed_smpl_params;
/* verts (F,6890,3), posed joints (F,24,3); scratch_A: F*24*12 floats (skinning transforms A, also the backward's
 * saved state); v_posed (F,6890,3) optional out: the un-skinned posed vertices, kept for maed_smpl_skin_bwd */
int maed_smpl_lbs_fwd(const maed_smpl_params* sp, const float* betas, const float* rotmat,
                      float* verts, float* joints24, float* scratch_A, float* v_posed, int F, void* stream);
/* out[f][j][:] = sum_v Jreg[j][v] verts[f][v][:]  (J <= 32 rows; MFMA f32 32x32x2) */
int maed_joint_regress_fwd(const float* Jreg, int J, const float* verts, float* out, int F, void* stream);
/* ... through the regressor's non-zeros (CSR: rowptr (J+1), cols / vals (nnz), vertex ids ascending per row; J <= 64): SMPL's joint regressors are sparse */
int maed_joint_regress_csr_fwd(const int32_t* rowptr, const int32_t* cols, const float* vals, int J, const float* verts, float* out, int F, void* stream);
/* joints49 = gather(cat(joints24, verts[extra_vertex_ids(21)], extra9), joint_map(49 int64)) -- the
 * integer index work is bit-exact (smpl.py:98-99); kp2d = projection(joints, cam) (spin.py:113-157).
 * If joints_override (F,Jo,3) is non-NULL it replaces the 49 joints before projection (ktd.py:108-112). */
int maed_smpl_joints_project_fwd(const float* joints24, const float* verts, const int64_t* extra_vertex_ids,
                                 const float* extra9, const int64_t* joint_map, const float* cam,
                                 const float* joints_override, int Jo, float* kp3d, float* kp2d, int F, void* stream);

/* ---- decoder tail, BACKWARD (the training graph of ktd.py:69-124; fp32 throughout) ------------------------------
 * Order of calls for one step, given the gradients of the five outputs of KTD.get_output:
 *   joints_project_bwd -> smpl_skin_bwd -> [dvp (F,20670) x PS^T (217,20670) GEMM by the caller] -> smpl_chain_bwd
 *   -> rot6d_pose_bwd -> ktd_chain_bwd -> [two GEMMs by the caller] -> ktd_unpack_add.                            */

/* K14/K15 backward: gradient of (kp3d, kp2d) w.r.t. the 54 source joints (scatter-ADD through the integer joint_map,
 * duplicates summed in index order) and the camera.  kp3d = the forward's output.  d_kp3d / d_kp2d may be NULL (= 0).
 * d_cam (F,3) = projection gradient + d_cam_in (optional, row stride cam_in_stride floats).                          */
int maed_smpl_joints_project_bwd(const float* kp3d, const float* cam, const int64_t* joint_map, const float* d_kp3d,
                                 const float* d_kp2d, const float* d_cam_in, int64_t cam_in_stride, float* d_joints24,
                                 float* d_extra21, float* d_extra9, float* d_cam, int F, void* stream);
/* K12 backward, vertex part.  d_v = d_verts (optional) + scatter(d_extra21 at extra_vertex_ids) + Jextra^T d_extra9;
 * writes d_vposed (F,20670) = T_v^T d_v and ACCUMULATES (+=, atomics; caller zeroes) dA (F,24,12) = sum_v w_vj d_v [v_posed,1]^T */
int maed_smpl_skin_bwd(const maed_smpl_params* sp, const float* A, const float* v_posed, const float* d_verts,
                       const float* d_extra21, const int64_t* extra_vertex_ids, const float* d_extra9, const float* Jextra,
                       float* d_vposed, float* dA, int F, void* stream);
/* ... for the case d_verts == NULL (the training objective reads key points only): d_v is non-zero only on the vertices the extra joints read -- `active`
 * (n_active sorted vertex ids: the non-zero columns of Jextra and extra_vertex_ids; the host derives the list once per model).  Writes the COMPACT
 * d_vposed_active (F, n_active, 3); the pose-corrective GEMM then runs on the matching columns of [posedirs; shapedirs^T].  dA as above. */
int maed_smpl_skin_bwd_sparse(const maed_smpl_params* sp, const float* A, const float* v_posed, const int32_t* active, int n_active,
                              const float* d_extra21, const int64_t* extra_vertex_ids, const float* d_extra9, const float* Jextra,
                              float* d_vposed_active, float* dA, int F, void* stream);
/* K12 backward, kinematic chain.  dpf_dbeta (F,217): columns 0..206 = d pose_feature (posedirs . d_vposed), 207..216 =
 * shapedirs^T d_vposed (both from the caller's GEMM).  d_rotmat_in (F,24,9) / d_betas_in (row stride given) optional
 * upstream gradients.  Outputs d_rotmat (F,24,9), d_betas (F,10).                                                    */
int maed_smpl_chain_bwd(const maed_smpl_params* sp, const float* betas, const float* rotmat, const float* dA,
                        const float* d_joints24, const float* dpf_dbeta, const float* d_rotmat_in, const float* d_betas_in,
                        int64_t betas_in_stride, float* d_rotmat, float* d_betas, int F, void* stream);
/* K11 backward: d_pose6d (n,6) from d_rotmat (n,9) and d_angle_axis (row n at d_aa + (n/24)*aa_stride + (n%24)*3; optional) */
int maed_rot6d_pose_bwd(const float* pose6d, const float* d_rotmat, const float* d_aa, int64_t aa_stride,
                        float* d_pose6d, int64_t n_joints, void* stream);
/* K10 backward: d_out (F, ld_out>=157): [:, :144] = d_base (chain transposed), [:, 144:154] = d_shape, [:, 154:157] = d_cam;
 * d_w_anc (3420, overwritten) and d_b_feat (157, overwritten = column sums of d_out).                                 */
int maed_ktd_chain_bwd(const float* pose, const float* w_anc, const float* d_pose, const float* d_shape, const float* d_cam,
                       float* d_out, int64_t ld_out, float* d_w_anc, float* d_b_feat, int F, void* stream);
/* KTD regressor weights <-> the packed operands of the head GEMM (ktd.py:58-67: joint_regs.{0..23}, decshape, deccam).
 * w[j] is the (6, hidden + 6*n_anc(j)) weight of regressor j (j = 24: decshape (10,hidden), 25: deccam (3,hidden)).
 * pack: w_feat (157,hidden), b_feat (157), w_anc (3420).  unpack_add: the reverse, ACCUMULATING (+=) into gw/gb.      */
#define MAED_KTD_W_ANC 3420
typedef struct { const float* w[26]; const float* b[26]; float* gw[26]; float* gb[26]; } maed_ktd_ptrs;
int maed_ktd_pack(const maed_ktd_ptrs* t, int hidden, float* w_feat, float* b_feat, float* w_anc, void* stream);
int maed_ktd_unpack_add(const maed_ktd_ptrs* t, int hidden, const float* d_w_feat, const float* d_b_feat, const float* d_w_anc, void* stream);

/* ---- lib/core/loss.py LossVideo / LossImage as ONE fused forward+backward ------------------------
 * Frames are flattened: M2 frames carry 2D keypoints, the LAST M3 <= M2 of them also 3D keypoints and SMPL parameters
 * (loss.py:165-181: preds[sample_2d_count:]).  pred_kp3d / pred_theta / gradients point at frame M2-M3.
 * losses[8] (device): 0 kp_2d, 1 kp_3d, 2 shape, 3 pose, 4 norm (each already weighted), 5 total, 6 n_valid, 7 unused.
 * Gradients of the TOTAL w.r.t. the predictions are written to d_kp2d (M2,49,2), d_kp3d (M3,49,3), d_theta (M3,85).
 * partials: scratch of max(M2,M3)*8 doubles.  gt_kp3d may be NULL (LossImage without 3D labels: loss.py:266,279).     */
typedef struct { float w_kp2d, w_kp3d, w_pose, w_shape, w_norm; } maed_loss_weights;
int maed_loss_fwd_bwd(const float* pred_kp2d, const float* gt_kp2d, int M2, const float* pred_kp3d, const float* gt_kp3d,
                      const float* pred_theta, const float* gt_theta, const uint8_t* w_smpl, int M3,
                      const maed_loss_weights* w, float* losses, float* d_kp2d, float* d_kp3d, float* d_theta,
                      double* partials, void* stream);

/* acceleration term of LossVideo (loss.py:94-117; e_smpl_accl_loss > 0): pred_kp3d (N,T,49,3), gt_kp3d (N,T,49,4) whole clips, T >= 3.
 * loss[1] (device, fp64) = weight * mean over (N, T-2, 49, 3) of (conf^4 * (second difference of pred - of gt))^2;
 * d_kp3d (N,T,49,3) = its gradient w.r.t. pred_kp3d (overwritten). */
int maed_loss_accl_fwd_bwd(const float* pred_kp3d, const float* gt_kp3d, int N, int T, float weight, double* loss, float* d_kp3d, void* stream);

/* ---- backbone helpers (the convolutions themselves ride on MIOpen) ---------------------------------- */
/* StdConv2dSame weight standardisation (resnetv2.py:74-93) for ALL convolutions in one launch.
 * conv_table: device array of n_convs maed_ws_conv; filters are numbered globally (fstart).  Forward writes
 * (w - mean)/(std + eps) in the compute dtype, laid out (O, kh, kw, I) = channels_last, at dst_off of `out`,
 * and stats[2*f] = mean, stats[2*f+1] = 1/(std+eps).  Backward accumulates (+=) into each gw from gout. */
typedef struct {
    const float* w;     /* fp32 master weight (O, I, kh, kw) contiguous                      */
    float* gw;          /* fp32 gradient of w, same layout (backward; NULL in forward)       */
    const void* gout;   /* grad w.r.t. the standardised weight, (O, kh, kw, I); NULL = skip  */
    int64_t dst_off;    /* element offset of this convolution inside `out`                   */
    int64_t dst_t_off;  /* >= 0: forward also writes the transposed (kh*kw*I, O) image at this offset of `out`
                         * (B operand of the input-gradient GEMM when a 1x1 convolution runs on maed_gemm_nt); -1 = none */
    int32_t O, I, KHW, fstart;
    int32_t gout_f32;   /* backward: gout is fp32 (a maed_gemm_tn_wgrad result) instead of the compute dtype */
    int32_t pad_;
} maed_ws_conv;
int maed_weight_std_fwd(const void* conv_table, int n_convs, int n_filters, void* out, int dtype, float* stats, float eps, int transpose_tiles, void* stream);
/* transpose_tiles > 0: the number of 64 x 64 tiles of all convolutions with dst_t_off >= 0 (sum of ceil(O / 64) * ceil(I * KHW / 64): the table is device memory, the
 * caller counts) -- their transposed images are then written by a tile transpose of the forward image (128-byte runs) instead of element-wise strided stores;
 * 0: the one-kernel form */
int maed_weight_std_bwd(const void* conv_table, int n_convs, int n_filters, int dtype, const float* stats, float eps, void* stream);
/* GroupNorm(32 groups)(+ residual)(+ ReLU) on channels_last activations x (N, HW, C) (resnetv2.py:35-49,189-204):
 * y = act(GN(x) * gamma + beta [+ residual]).  sums: (N,32,2) doubles written by forward, read by backward.
 * sums_zeroed / ab_zeroed != 0: the caller hands in scratch that is already zero (one memset for all 52 layers of a
 * backbone pass instead of one per layer).  sums_zeroed == 2: sums already HOLD the statistics of x -- accumulated by the producing
 * convolution's epilogue (maed_conv1x1_fwd / maed_conv3x3_fwd gn_sums) -- and the statistics pass over x is skipped. */
int maed_groupnorm_fwd(const void* x, const void* residual, const float* gamma, const float* beta, void* y, double* sums,
                       uint8_t* relu_mask, int N, int HW, int C, float eps, int relu, int dtype, int sums_zeroed, void* stream);
/* maed_groupnorm_fwd on fp32 tensors that ALSO writes bf16 twins of its input (twin_x, may be NULL) and of its result (twin_y, may be NULL), same shapes: what a
 * bf16 backward reads behind an fp32 forward ("bf16x3 forward / bf16 backward from bf16 twins", maed_amd/resnetv2.py).  resnetv2.py:35-49. */
int maed_groupnorm_fwd_twin(const void* x, const void* residual, const float* gamma, const float* beta, void* y, double* sums,
                            uint8_t* relu_mask, int N, int HW, int C, float eps, int relu, int sums_zeroed, void* twin_x, void* twin_y, void* stream);
/* relu_mask (N*HW*C/8 bytes, bit j of byte (n,hw,c/8) = output channel 8*(c/8)+j > 0): written by forward when a residual is
 * added before the ReLU (optional: inference passes NULL), required by backward in that case -- without a residual the mask is
 * recomputed from x.  16x less traffic than re-reading the saved output in both backward passes.
 * dx (and dres = masked dy when dres != NULL; with a relu_mask dres may be NULL: the consumer masks dy itself, MAED_EPI_ADD with out2 = relu_mask);
 * dgamma/dbeta += (atomics); ab_scratch: N*C*2 floats.
 * frame_sync (optional, N * MAED_GN_SYNC_WORDS 4-byte words that are ZERO at launch -- e.g. a slice of the same zero-filled arena as ab_scratch; single use):
 * with it the backward reads x and dy ONCE -- the workgroups that share a frame keep their slices in registers between the reduction and the apply step,
 * exchange their 2 x 32 group sums through frame_sync[n] and meet at its arrival counter (3 tensor streams instead of 5; MAED_OPT_GN_BWD_ONEPASS = 0 or
 * NULL: the two-pass kernels) */
#define MAED_GN_SYNC_WORDS 80
int maed_groupnorm_bwd(const void* x, const uint8_t* relu_mask, const void* dy, const double* sums, const float* gamma, const float* beta,
                       void* dx, void* dres, float* dgamma, float* dbeta, float* ab_scratch, int N, int HW, int C, float eps,
                       int relu, int dtype, int ab_zeroed, uint32_t* frame_sync, void* aux_stream, void* stream);
/* dgamma == NULL and dbeta == NULL: the closing column sum over the frames is left to the caller -- ab_scratch then holds, per frame and channel, the (dbeta, dgamma)
 * partials of this layer, and ONE maed_gn_affine_grad_batch call folds the partials of many layers (a backbone pass: 52) into their gradients: one launch instead of
 * one per layer (round 6: 52 x 6 us of launch granularity per step).  items: HOST array; dgamma[c] += sum_n ab[n][c][1], dbeta[c] += sum_n ab[n][c][0]. */
typedef struct { const float* ab; float* dgamma; float* dbeta; int N; int C; } maed_gn_affine_item;
int maed_gn_affine_grad_batch(const maed_gn_affine_item* items, int count, void* stream);
/* aux_stream (optional): a second stream of the caller's on which the closing dgamma/dbeta column sum is enqueued (fenced after the reduction
 * pass on `stream`); the caller joins it before anybody reads dgamma/dbeta.  NULL: everything on `stream`. */

/* MaxPool2dSame(kernel 3, stride 2) of the stem (resnetv2.py:61-72) on channels_last x (N,H,W,C), C % 8 == 0: y (N,ceil(H/2),
 * ceil(W/2),C) and the winning tap per output element (idx, uint8, same shape as y; ATen tie/NaN rule); backward gathers dx. */
int maed_maxpool3s2_same_fwd(const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, int dtype, void* stream);
int maed_maxpool3s2_same_bwd(const void* dy, const uint8_t* idx, void* dx, int N, int H, int W, int C, int dtype, void* stream);

/* Input of the stem convolution in one pass: x fp32 (N,C,H,W) contiguous -> y (N, H + pad_top + pad_bottom, W + pad_left + pad_right, c_stride) channels_last in
 * the compute dtype, zero borders (the TF-SAME padding of resnetv2.py:51-59 materialised), channel slots C .. c_stride-1 zero.  C <= c_stride <= 4
 * (c_stride 4: 8-byte pixels for maed_stem7x7s2_*; y 16-B aligned). */
int maed_stem_input(const float* x, void* y, int N, int C, int H, int W, int pad_top, int pad_bottom, int pad_left, int pad_right, int c_stride, int dtype,
                    void* stream);

/* The stem convolution itself -- StdConv2dSame(3 -> 64, kernel 7, stride 2), resnetv2.py:74-93 as instantiated by :330-333 -- on the library (csrc/stem.hip;
 * replaces the vendor convolution F.conv2d dispatches to).  bf16 only; even H, W with (H/2)*(W/2) % 128 == 0 and (W/2) % 16 == 0 (maed_stem7x7s2_supported).
 *   xp   (F, H+5, W+6, 4)  maed_stem_input(pad_top 2, pad_bottom 3, pad_left 2, pad_right 4, c_stride 4): TF-SAME padding plus one zero column (even width)
 *   w    (64, 7, 7, 3)     the standardised weight, channels_last (O, kh, kw, I) memory order
 *   wimg 28672 bytes of scratch (the fragment-major weight image the forward builds for itself), 16-B aligned
 *   y    (F, H/2, W/2, 64) channels_last;  gn_sums (optional): (F, 32, 2) fp64 sum / sum of squares of the rounded outputs per 2-channel group, ACCUMULATED
 *        (the statistics of the GroupNorm behind the stem, as maed_conv3x3_fwd / maed_conv1x1_fwd produce them)
 *   dW   (64, 7, 7, 3) fp32, ACCUMULATED (zero it once per step); scratch (optional): maed_stem7x7s2_wgrad_scratch_floats(F, H, W) fp32 elements for the
 *        per-workgroup partial results (plain stores + one reduction pass); NULL: the workgroups add into dW with atomics (slower: one 37 KB hot spot)
 * Non-finite pixels: the zero-weight slots (kx = 7, c = 3) are multiplied like any other, so an Inf/NaN pixel also reaches the one output column to its left. */
/* Weight gradient of the stride-1 3x3 SAME convolution at 64 -> 64 channels (the conv2 of the stage-1 bottlenecks, resnetv2.py:218-233) one image row per work item
 * (csrc/conv3x3_rows.hip); maed_conv3x3_wgrad takes the same kernel for this shape with atomics.  dy, x (F, H, W, 64) channels_last bf16; W % 8 == 0, 8 <= W <= 64.
 * dW (64, 3, 3, 64) fp32, ACCUMULATED.  scratch: maed_conv3x3_wgrad_rows64_scratch_floats(...) fp32 elements (0 = shape not covered) for per-workgroup partial results --
 * NULL: the workgroups add into dW with atomics (slower: every workgroup adds onto the same 147 KB). */
int maed_conv3x3_wgrad_rows64_scratch_floats(int F, int H, int W, int Cin, int Cout);
int maed_conv3x3_wgrad_rows64(const void* dy, const void* x, float* dW, void* scratch, int F, int H, int W, int dtype, void* stream);

int maed_stem7x7s2_supported(int H, int W);
int maed_stem7x7s2_fwd(const void* xp, const void* w, void* wimg, void* y, double* gn_sums, int F, int H, int W, int dtype, void* stream);
int maed_stem7x7s2_wgrad_scratch_floats(int F, int H, int W);
int maed_stem7x7s2_wgrad(const void* dy, const void* xp, float* dW, void* scratch, int F, int H, int W, int dtype, void* stream);

/* Pixel subsampling of the 1x1 stride-2 downsample convolutions (resnetv2.py:207-216; TF-SAME padding of a 1x1 kernel is always zero):
 * fwd: y (F,ceil(H/2),ceil(W/2),C) = x[:, ::2, ::2, :] on channels_last x (F,H,W,C), C % 8 == 0 -- the convolution is then
 * maed_conv1x1_fwd on the packed rows; bwd: dx (F,H,W,C) = the packed gradient g spread back, zeros elsewhere (one write pass). */
int maed_subsample2_fwd(const void* x, void* y, int F, int H, int W, int C, int dtype, void* stream);
int maed_subsample2_bwd(const void* g, void* dx, int F, int H, int W, int C, int dtype, void* stream);

/* compute-dtype images of the nn.Linear master weights after an optimizer step (ops.WeightCache), all in one launch:
 * for every entry dst_c (rows, cols) = cast(src) (NULL = skip) and dst_t (cols, rows) = cast(src)^T.
 * table: device array of n_entries maed_wt_entry; tile0 = index of the entry's first 64x64 tile, tiles_n = ceil(cols/64);
 * n_tiles = total tile count = launch grid. */
typedef struct { const float* src; void* dst_c; void* dst_t; int32_t rows, cols, tile0, tiles_n; } maed_wt_entry;
int maed_weight_refresh(const void* table, int n_entries, int n_tiles, int dtype, void* stream);

/* ---- optimizer: Adam (lib/utils/utils.py:127-132; torch.optim.Adam semantics, L2 weight decay) - */
/* flat fp32 arenas p,g,m,v of n elements; grad is scaled by gscale first (1/world for DDP mean).
 * Optionally refreshes the bf16 shadow copy of the parameters (shadow_bf16 may be NULL). */
int maed_adam_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay,
                   float bias_corr1, float bias_corr2, float gscale, void* stream);
/* the same with lr, bias_corr1, bias_corr2 read from the device record `state` when the kernel runs (see maed_train_state) */
int maed_adam_step_dev(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, const maed_train_state* state,
                       float beta1, float beta2, float eps, float weight_decay, float gscale, void* stream);

/* ---- gradient all-reduce: RCCL over xGMI on a side HIP stream (train.py:113,182 DDP/NCCL) ----------------------- */
/* One communicator per process.  RCCL is bound at run time from the library the host names (NULL = "librccl.so.1");
 * pass the copy PyTorch-ROCm already loaded so the process holds one RCCL.  unique_id: MAED_COMM_ID_BYTES made by
 * maed_comm_unique_id on rank 0 and distributed by the caller (the host uses the torch.distributed store).
 * allreduce_async: SUM in place over n elements; ordered after everything already enqueued on compute_stream, runs on the
 * library's side stream (overlaps later kernels of compute_stream).  wait: compute_stream waits for all issued collectives. */
#define MAED_COMM_ID_BYTES 128
int maed_comm_load(const char* librccl_path);
int maed_comm_unique_id(void* id128);
int maed_comm_init(int rank, int world, const void* unique_id);
int maed_comm_allreduce_async(void* buf, size_t n, int dtype, void* compute_stream);
int maed_comm_wait(void* compute_stream);
int maed_comm_world(void);
int maed_comm_destroy(void);

/* ---- 3x3 convolution as an implicit GEMM (resnetv2.py:74-93 StdConv2dSame, kernel 3; bf16, channels_last) -----------------
 * y (F,Ho,Wo,Cout) = conv(x (F,H,W,Cin), w) with TF-SAME zero padding (pad_top / pad_left rows / columns in front; the rest behind),
 * any stride; w_taps is the weight as (Cout, 3, 3, Cin) = (Cout, 9*Cin) row-major (maed_weight_std_fwd's output order).
 * zero_page: >= 128 bytes of zeros in device memory (source of every out-of-image tap).  add (optional, (F,Ho,Wo,Cout)): y = conv + add.
 * The input gradient of a stride-1 convolution is this entry point on dY (Cin := forward Cout, Cout := forward Cin) with either the
 * flipped, transposed weight w'[ci][2-ky][2-kx][co] (w_layout 0) or, copy-free, the transposed image (3,3,Cin_f,Cout_f) that
 * maed_weight_std_fwd writes next to the forward image (w_layout 1: the tap flip becomes a negative tap stride).
 * Needs Cin % 64 == 0, Cout % 8 == 0.  The backbone's default 3x3 path (maed_amd/resnetv2.py; MAED_CONV3X3=miopen switches back).
 * gn_sums (optional, fp64 (F,32,2), PRE-ZEROED): the GroupNorm(32) statistics of the stored output -- sums[f][g] += (sum y, sum y^2)
 * over the pixels of frame f and the channels of group g -- for the GroupNorm that follows every convolution (resnetv2.py:35-49):
 * maed_groupnorm_fwd then skips its statistics pass (stats = 2).  Needs Cout = 32 * 2^k >= 64, Ho*Wo >= 128, add == NULL. */
int maed_conv3x3_fwd(const void* x, const void* w_taps, const void* zero_page, void* y, int F, int H, int W, int Cin, int Cout,
                     int stride, int pad_top, int pad_left, int Ho, int Wo, const void* add, int w_layout, int dtype, double* gn_sums,
                     void* stream);

/* 1x1 stride-1 convolution on a channels_last activation viewed as (M = F*H*W, Cin) rows: y (M, Cout) = x w^T, w (Cout, Cin) the
 * standardised weight (no bias: StdConv2dSame) -- maed_gemm_nt's STORE epilogue -- plus, optionally, the GroupNorm statistics of the
 * output exactly as maed_conv3x3_fwd's gn_sums (hw = H*W pixels per frame).  bf16, Cin % 64 == 0. */
int maed_conv1x1_fwd(const void* x, int64_t ldx, const void* w, int64_t ldw, int64_t M, int Cout, int Cin, void* y, int64_t ldy, int hw,
                     double* gn_sums, int dtype, void* stream);

/* Input gradient of a STRIDE-2 3x3 SAME convolution (conv2 of the first block of stages 2 and 3, resnetv2.py:74-93,159-204) on the library's implicit-GEMM kernel:
 * four launches, one per parity class of the input pixel (2 or 1 forward taps per axis and class; every pixel of dx is written exactly once: no zero-fill).
 * dy (F,Ho,Wo,Cout), dx (F,H,W,Cin) channels_last bf16; wt_image (3,3,Cin,Cout): the transposed image of the standardised forward weight as maed_weight_std_fwd
 * writes it; pad_top / pad_left: the forward's TF-SAME padding (0 or 1).  Cout % 64 == 0, Cin % 8 == 0. */
int maed_conv3x3_s2_dgrad(const void* dy, const void* wt_image, const void* zero_page, void* dx, int F, int H, int W, int Cin, int Cout,
                          int pad_top, int pad_left, int Ho, int Wo, int dtype, void* stream);
/* Weight gradient of the same stride-2 convolution: dW (Cout,3,3,Cin) fp32 += over the output pixels of dy (F,Ho,Wo,Cout) x the input rows of x (F,H,W,Cin) they see --
 * the TN weight-gradient kernel over gathered rows.  maed_conv3x3_s2_tables fills, once per feature-map geometry, the per-output-pixel tables it gathers through:
 * tapmask (uint16, bit t = tap (t/3, t%3) inside the image) and rowtab (int32: x row of the top-left tap), both padded to a multiple of 64 entries.
 * F*Ho*Wo % 64 == 0, Cin, Cout % 8 == 0, bf16. */
int maed_conv3x3_s2_tables(void* tapmask, int* rowtab, int F, int H, int W, int pad_top, int pad_left, int Ho, int Wo, void* stream);
int maed_conv3x3_s2_wgrad(const void* dy, const void* x, const void* tapmask, const int* rowtab, const void* zero_page, float* dW, int F, int H, int W,
                          int Cin, int Cout, int Ho, int Wo, int dtype, void* stream);
/* weight gradient of the stride-1 3x3 SAME convolution: dW (Cout, 9*Cin) fp32 += sum over pixels of dy (F,H,W,Cout) x shifted x (F,H,W,Cin)
 * (a TN GEMM over gathered rows on maed_gemm_tn_wgrad's kernel).  tapmask: F*H*W uint16 rounded up to a multiple of 64, filled once per
 * (F,H,W) by maed_conv3x3_tapmask; zero_page as for maed_conv3x3_fwd.  Needs F*H*W % 64 == 0, Cin % 8 == 0, Cout % 8 == 0. */
int maed_conv3x3_tapmask(void* tapmask, int F, int H, int W, void* stream);
int maed_conv3x3_wgrad(const void* dy, const void* x, const void* tapmask, const void* zero_page, float* dW, int F, int H, int W, int Cin,
                       int Cout, int dtype, void* stream);

/* ---- evaluation metrics on the device ----------------------------------------------------
 * Replace the numpy / torch-CPU post-processing of lib/core/evaluate.py:135-166 and lib/utils/eval_utils.py.  fp32 in/out.
 *
 * maed_eval_pose_errors: per frame n of pred (N,J,3) / target (N,J,4 = x,y,z,visibility), 4 <= J <= 64:
 *   both sides * visibility (evaluate.py:142-146), minus their pelvis = midpoint of joints 2 and 3 (:151-155);
 *   mpjpe[n]    = mean_j ||pred - target||                                   (:158)
 *   pa_mpjpe[n] = the same after the similarity transform (s, R, t) of eval_utils.py:201-252
 *                 (batch_compute_similarity_transform_torch; the 3x3 SVD is solved in-kernel)   (:159-160)
 *   pred_centred / target_centred (N,J,3), optional: the masked, centred joints (inputs of the acceleration metrics). */
int maed_eval_pose_errors(const float* pred, const float* target, int N, int J, float* mpjpe, float* pa_mpjpe,
                          float* pred_centred, float* target_centred, void* stream);
/* eval_utils.py:201-252 batch_compute_similarity_transform_torch on (N,J,3) point sets, J <= 64:
 * S1_hat[n] = s R S1[n] + t, the similarity transform of S1[n] closest to S2[n] (det R = +1). */
int maed_similarity_transform(const float* S1, const float* S2, int N, int J, float* S1_hat, void* stream);
/* out[n] = mean_j ||a[n] - 2 a[n+1] + a[n+2]||, n < N-2; a = joints (N,J,3) (eval_utils.py:10-21 compute_accel), or
 * joints - joints_gt when joints_gt != NULL (eval_utils.py:24-52 compute_error_accel with vis=None). */
int maed_eval_accel(const float* joints, const float* joints_gt, int N, int J, float* out, void* stream);
/* out[n] = mean_v ||pred_verts[n,v] - target_verts[n,v]||, (N,V,3) each (eval_utils.py:88-90 compute_error_verts). */
int maed_eval_vertex_error(const float* pred_verts, const float* target_verts, int N, int V, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MAED_HIP_H */
