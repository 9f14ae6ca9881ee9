// The stem convolution of the hybrid backbone: StdConv2dSame(3 -> 64, kernel 7, stride 2) on the clip frames (resnetv2.py:74-93 through :330-333, the first
// layer of every forward), forward and weight gradient (the frames need no input gradient).  bf16 operands, fp32 accumulation.
//
// The input is the pre-padded channels_last image of maed_stem_input with FOUR channels per pixel (the fourth is zero) and an even padded width:
//   xp (F, Hp, Wp, 4) bf16,  Hp = H + 5, Wp = W + 6 for even H, W (TF-SAME: 2 rows / columns before, 3 after, one more zero column to make Wp even)
// A pixel is then 8 bytes and the stride-2 step from one output pixel to the next is 16 bytes = one 8-element MFMA fragment:
//   patch(oy, ox)[ky][e] = xp_row(2 oy + ky)[8 ox + e],  e = 4 kx + c = 0 .. 31   (kx = 7 and c = 3 meet zero weights)
// so the reduction is laid out as K = 7 kernel rows x 32 elements = 224 (147 of them real), and
//   forward:  the fragment "8 consecutive k of output pixel ox" is ONE aligned 16-byte load, consecutive lanes (ox) read consecutive 16-byte chunks -- the
//             pixel operand goes from global memory straight into MFMA registers (no im2col, no LDS staging); only the 28 KB weight image lives in LDS;
//   wgrad:    dW[co][ky][e] = sum_ox dy[ox][co] * xp_row[8 ox + e] contracts over PIXELS: both operands are needed pixel-contiguous per lane.  The output row of
//             dy (Wo x 64) and the 7 input rows are copied unchanged into LDS by LDS-DMA and read back transposed with ds_read_b64_tr_b16: for the image operand
//             the "matrix" [pixel][e] is the raw row with a row stride of 16 bytes -- the im2col overlap costs nothing.
#include "common.cuh"
#include "prof.h"

#define STEM_K 224                  // padded reduction length: 7 x 32
#define STEM_CO 64
#define STEM_WIMG (STEM_CO * STEM_K)

// fragment-major weight image: img[((kk * 2 + hi) * 64 + co) * 8 + j] = w[co][ky][kx][c],  k = 16 kk + 8 hi + j = 32 ky + 4 kx + c  (zero for kx = 7 or c = 3):
// the A fragment of (k-step kk, half hi) is 64 consecutive 16-byte slots -- a conflict-free ds_read_b128 with lane = co
__global__ __launch_bounds__(256) void stem_wimg_kernel(const bf16* __restrict__ w, bf16* __restrict__ img) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= STEM_WIMG) return;
    const int j = idx & 7, co = (idx >> 3) & 63, hi = (idx >> 9) & 1, kk = idx >> 10;
    const int k = 16 * kk + 8 * hi + j, ky = k >> 5, e = k & 31, kx = e >> 2, c = e & 3;
    img[idx].v = (kx < 7 && c < 3) ? w[((co * 7 + ky) * 7 + kx) * 3 + c].v : (unsigned short)0;
}

// One workgroup: 4 waves x tpw tiles of 32 consecutive output pixels (all inside one frame: Ho*Wo % (128 tpw) == 0, host-checked), all 64 output channels.
// D = W (rows: channel) x pixels (columns): a lane ends up with 4 consecutive channels of ITS pixel per register quad -> 8-byte channels_last stores.
// GN: the GroupNorm statistics of the layer behind (32 groups of 2 channels; sum and sum of squares of the ROUNDED outputs) come from the matrix cores too: the
// wave's output tile sits in LDS as [pixel][channel] on its way out, ds_read_b64_tr_b16 hands it back as fragments with lane = channel, k = pixel, and per 16
// channels one v_mfma_f32_16x16x32_bf16 with a ones operand gives the column sums, one with the fragment on both sides the Gram matrix whose diagonal is the sum
// of squares (bf16 products are exact in fp32): 8 small MFMAs + 8 transposing reads per tile and no VALU work in the loop.  (Per-lane VALU accumulation -- plain,
// or v_dot2c_f32_bf16 on the packed pairs -- cost 17-33 us of a 68 us launch: the loop's VALU slots are what the epilogue competes for, the MFMA pipe is idle.)
// (The 28 weight fragments in registers for the whole workgroup -- 112 VGPRs, one wave per SIMD -- measured 97 against 66 us: re-read from LDS per tile.)
template <bool GN>
__global__ __launch_bounds__(256, 3) void stem7x7s2_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ wimg, bf16* __restrict__ y,
                                                               double* __restrict__ gn_sums, int Hp, int Wp, int Ho, int Wo, int tpw) {
    __shared__ __attribute__((aligned(16))) unsigned short ws[STEM_WIMG];
    __shared__ __attribute__((aligned(16))) unsigned short otile[4 * 32 * 64];      // per wave: one output tile (32 pixels x 64 channels) on its way to full-line stores
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    for (int i = tid; i < STEM_WIMG / 8; i += 256) reinterpret_cast<uint4*>(ws)[i] = reinterpret_cast<const uint4*>(wimg)[i];
    __syncthreads();
    const int hw = Ho * Wo;
    const int64_t p_wg = (int64_t)blockIdx.x * (128 * tpw);
    const int f = (int)(p_wg / hw);
    const int q_wave = (int)(p_wg - (int64_t)f * hw) + wave * tpw * 32;          // first in-frame pixel of this wave
    const int row_elems = Wp * 4;
    f32x4_t ssum[4], sgram[4];                     // per block of 16 channels: column sums (every row the same), Gram matrix
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[b][r] = 0.f; sgram[b][r] = 0.f; }
    // transposing reads of the output tile: 16-lane group q covers pixels 8 q .. 8 q + 7 (two reads of 4), lane i16 of it pixel row (i16 >> 2), channels 4 (i16 & 3) ..
    const int i16 = lane & 15, srow = 8 * (lane >> 4) + (i16 >> 2), scol = 4 * (i16 & 3);

    uint4 cur[14];
#define STEM_LOAD(dst_, t_) { const int q__ = q_wave + (t_) * 32 + l31; const int oy__ = q__ / Wo, ox__ = q__ - oy__ * Wo; \
        const bf16* b__ = x + (((int64_t)f * Hp + 2 * oy__) * Wp + 2 * ox__) * 4 + hi * 8; \
        _Pragma("unroll") for (int ky = 0; ky < 7; ++ky) { dst_[2 * ky] = *reinterpret_cast<const uint4*>(b__ + ky * row_elems); \
                                                            dst_[2 * ky + 1] = *reinterpret_cast<const uint4*>(b__ + ky * row_elems + 16); } }
    STEM_LOAD(cur, 0);
    for (int t = 0; t < tpw; ++t) {
        f32x16_t acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
        asm volatile("" ::: "memory");        // keep the weight-fragment reads inside the tile loop (hoisted they cost 112 VGPRs)
#pragma unroll
        for (int kk = 0; kk < 14; ++kk) {
            const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(ws + ((kk * 2 + hi) * 64 + l31) * 8);
            const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(ws + ((kk * 2 + hi) * 64 + 32 + l31) * 8);
            union { uint4 u; bf16x8_t v; } b;
            b.u = cur[kk];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b.v, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b.v, acc[1], 0, 0, 0);
        }
        // the next tile's pixel fragments are requested as soon as this tile's MFMAs have taken theirs: the epilogue below covers the latency (a second register
        // set loaded one tile ahead cost 56 VGPRs and a resident wave per SIMD)
        if (t + 1 < tpw) STEM_LOAD(cur, t + 1);
        // acc[a][4 g + i] = channel 32 a + 8 g + 4 hi + i of pixel l31 of the tile.  Straight from here a store instruction would touch 32 output lines with 16 bytes
        // each (12.9 M partial-line writes per launch: the first version's limiter); through the wave's 4 KB LDS patch -- 16-byte chunk c of pixel p at slot
        // c ^ (p & 7): conflict-free ds_write_b64 and ds_read_b128 -- every store instruction writes one contiguous KB.  (LDS operations of one wave execute in order.)
        unsigned short* ot = otile + wave * (32 * 64);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint32_t u0 = pack_bf2(acc[a][4 * g], acc[a][4 * g + 1]), u1 = pack_bf2(acc[a][4 * g + 2], acc[a][4 * g + 3]);
                *reinterpret_cast<uint2*>(ot + l31 * 64 + (((4 * a + g) ^ (l31 & 7)) << 3) + 4 * hi) = make_uint2(u0, u1);
            }
        MAED_WAVE_LDS_SYNC();
        if constexpr (GN) {
            union { bf16x8_t v; uint32_t u[4]; } ones;
            ones.u[0] = ones.u[1] = ones.u[2] = ones.u[3] = 0x3f803f80u;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int col = 16 * b + scol;
                union { bf16x8_t v; uint2 u[2]; } fr;
                auto lo = MAED_DS_READ_TR16(ot + srow * 64 + (((col >> 3) ^ (srow & 7)) << 3) + (col & 7));
                auto h2 = MAED_DS_READ_TR16(ot + (srow + 4) * 64 + (((col >> 3) ^ ((srow + 4) & 7)) << 3) + (col & 7));
                __builtin_memcpy(&fr.u[0], &lo, 8); __builtin_memcpy(&fr.u[1], &h2, 8);
                ssum[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones.v, fr.v, ssum[b], 0, 0, 0);
                sgram[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr.v, fr.v, sgram[b], 0, 0, 0);
            }
        }
        uint4* yt = reinterpret_cast<uint4*>(y + (p_wg + (int64_t)(wave * tpw + t) * 32) * STEM_CO);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = r * 64 + lane, p = n >> 3, c = n & 7;
            yt[n] = *reinterpret_cast<const uint4*>(ot + p * 64 + ((c ^ (p & 7)) << 3));
        }
        MAED_WAVE_LDS_SYNC();
    }
#undef STEM_LOAD
    if constexpr (GN) {
        // D[i][j] of the 16x16 products: column j = lane & 15, rows 4 (lane >> 4) + r: the column sum is in every row, the diagonal element of channel j in the
        // 16-lane group j >> 2, register j & 3
        float* part = reinterpret_cast<float*>(otile);                 // [4 waves][sum | sum of squares][64 channels]  (the tiles are out: every wave synced after its reads)
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (lane < 16) part[(wave * 2) * 64 + 16 * b + i16] = ssum[b][0];
            if ((lane >> 4) == (i16 >> 2)) {
                const int r = i16 & 3;
                part[(wave * 2 + 1) * 64 + 16 * b + i16] = r == 0 ? sgram[b][0] : r == 1 ? sgram[b][1] : r == 2 ? sgram[b][2] : sgram[b][3];
            }
        }
        __syncthreads();
        if (tid < 64) {
            const int grp = tid >> 1, k = tid & 1;
            double t4 = 0.0;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) t4 += (double)part[(wv * 2 + k) * 64 + 2 * grp] + (double)part[(wv * 2 + k) * 64 + 2 * grp + 1];
            atomicAdd(gn_sums + (int64_t)f * 64 + tid, t4);
        }
    }
}

// ---- weight gradient ---------------------------------------------------------------------------------------------------------------------------------
// One work item = one output row (f, oy): dy row (Wo pixels x 64 channels = Wo * 8 chunks of 16 bytes, contiguous) and input rows 2 oy .. 2 oy + 6 (7 * Wp / 2
// chunks, contiguous).  Both are copied into LDS by LDS-DMA (448 lanes x 16 bytes per round; the last round's spare lanes re-load the last chunk into slack
// space), double-buffered: the copy of item i + 1 runs under the MFMAs of item i, one barrier per item.  Wave ky (7 waves) owns kernel row ky: per 16-pixel
// k-step two A fragments (channels 0-31 / 32-63 of the dy row) and one B fragment (the 32 elements of its kernel row), all ds_read_b64_tr_b16 pairs of the
// row-major images, two MFMAs.  The dy image is stored with 16-byte chunk c of pixel row r at chunk c ^ 4 * ((r >> 1) & 1): pixel rows are 128 bytes apart, and
// a transposing read covers 4 consecutive pixels -- unswizzled, pixels r and r + 2 would meet on the same banks.  (The image operand's row stride is 16 bytes:
// a whole wave's read spans < 256 bytes.)  A workgroup walks a contiguous range of rows (consecutive rows share 5 of their 7 input rows: L2 hits) and adds its
// 64 x 147 partial gradient with fp32 atomics.
#define STEM_WG_THREADS 448
__global__ __launch_bounds__(STEM_WG_THREADS, 2) void stem7x7s2_wgrad_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x, float* __restrict__ dW,
                                                                              float* __restrict__ partial, int Hp, int Wp, int Ho, int Wo, int n_items, int items_per_wg) {
    MAED_DYN_SHARED(unsigned short, smem);
    const int tid = threadIdx.x, lane = tid & 63, ky = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5, i16 = lane & 15;
    const int dy_chunks = Wo * 8, x_chunks = 7 * (Wp / 2);
    const int dy_rounds = (dy_chunks + STEM_WG_THREADS - 1) / STEM_WG_THREADS, x_rounds = (x_chunks + STEM_WG_THREADS - 1) / STEM_WG_THREADS;
    const int buf_elems = (dy_rounds + x_rounds) * STEM_WG_THREADS * 8;
    const int item0 = blockIdx.x * items_per_wg;
    int item1 = item0 + items_per_wg;
    if (item1 > n_items) item1 = n_items;
    if (item0 >= item1) return;

#define STEM_ISSUE(item_, b_) { const int it__ = (item_); const int f__ = it__ / Ho, oy__ = it__ - f__ * Ho; \
        const char* dyb__ = (const char*)dy + (int64_t)it__ * dy_chunks * 16; \
        const char* xb__ = (const char*)x + ((int64_t)f__ * Hp + 2 * oy__) * Wp * 8; \
        unsigned short* lb__ = smem + (size_t)(b_) * buf_elems; \
        for (int j = 0; j < dy_rounds; ++j) { int p = j * STEM_WG_THREADS + tid; if (p > dy_chunks - 1) p = dy_chunks - 1; const int row = p >> 3, c = p & 7; \
            MAED_LDS_DMA16(dyb__, (uint32_t)((row * 8 + (c ^ (((row >> 1) & 1) << 2))) * 16), lb__ + (j * STEM_WG_THREADS + ky * 64) * 8); } \
        for (int j = 0; j < x_rounds; ++j) { int p = j * STEM_WG_THREADS + tid; if (p > x_chunks - 1) p = x_chunks - 1; \
            MAED_LDS_DMA16(xb__, (uint32_t)(p * 16), lb__ + ((dy_rounds + j) * STEM_WG_THREADS + ky * 64) * 8); } }

    f32x16_t acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    // lane constants of the transposing reads: pixel row 4 hi + (i16 >> 2) (+ 8 for the second read) of a k-step, columns (lane & 16) + 4 (i16 & 3) ...
    const int prow = 4 * hi + (i16 >> 2), pcol = (lane & 16) + 4 * (i16 & 3);
    const int sw = (i16 >> 3) << 2;                                   // ((pixel row >> 1) & 1) << 2: the same for row, row + 8 and every k-step
    const int a_off0 = prow * 64 + ((((pcol >> 3) ^ sw) << 3) | (pcol & 7)), a_off1 = prow * 64 + (((((pcol + 32) >> 3) ^ sw) << 3) | (pcol & 7));
    const int b_off = ky * Wp * 4 + prow * 8 + pcol;
    const int ksteps = Wo >> 4;

    STEM_ISSUE(item0, 0);
    for (int item = item0; item < item1; ++item) {
        const int b = (item - item0) & 1;
        MAED_WAIT_VMCNT0();
        __syncthreads();                 // item's images have landed (every thread's DMAs) and nobody still reads the other buffer
        if (item + 1 < item1) STEM_ISSUE(item + 1, b ^ 1);
        const unsigned short* dys = smem + (size_t)b * buf_elems;
        const unsigned short* xs = dys + dy_rounds * STEM_WG_THREADS * 8;
        for (int s = 0; s < ksteps; ++s) {
            union { bf16x8_t v; uint2 u[2]; } a0, a1, bb;
            { auto lo = MAED_DS_READ_TR16(dys + s * 16 * 64 + a_off0); auto h2 = MAED_DS_READ_TR16(dys + s * 16 * 64 + 8 * 64 + a_off0);
              __builtin_memcpy(&a0.u[0], &lo, 8); __builtin_memcpy(&a0.u[1], &h2, 8); }
            { auto lo = MAED_DS_READ_TR16(dys + s * 16 * 64 + a_off1); auto h2 = MAED_DS_READ_TR16(dys + s * 16 * 64 + 8 * 64 + a_off1);
              __builtin_memcpy(&a1.u[0], &lo, 8); __builtin_memcpy(&a1.u[1], &h2, 8); }
            { auto lo = MAED_DS_READ_TR16(xs + s * 16 * 8 + b_off); auto h2 = MAED_DS_READ_TR16(xs + s * 16 * 8 + 8 * 8 + b_off);
              __builtin_memcpy(&bb.u[0], &lo, 8); __builtin_memcpy(&bb.u[1], &h2, 8); }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.v, bb.v, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, bb.v, acc[1], 0, 0, 0);
        }
    }
#undef STEM_ISSUE
    // D[channel][e]: column e = l31 of kernel row ky, rows (r & 3) + 8 (r >> 2) + 4 hi
    const int e = l31, kx = e >> 2, c = e & 3;
    // with `partial`: plain stores into the workgroup's own 64 x 147 slot, summed by wgrad_slots_reduce_kernel (conv3x3_rows.hip) -- 512 workgroups adding onto the
    // same 37 KB with atomics is a hot spot
    if (kx < 7 && c < 3) {
        float* d = (partial ? partial + (size_t)blockIdx.x * (64 * 147) : dW) + (ky * 7 + kx) * 3 + c;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* e2 = d + (32 * a + (r & 3) + 8 * (r >> 2) + 4 * hi) * 147;
                if (partial) *e2 = acc[a][r]; else atomicAdd(e2, acc[a][r]);
            }
    }
}

void maed_wgrad_slots_reduce(const float* partial, float* dW, int n_slots, int n_elems, hipStream_t stream);      // conv3x3_rows.hip

static int stem_wgrad_wgs(int n_items, int* per_out) {
    int wgs = maed_opt(MAED_OPT_STEM_WGRAD_WGS);          // default 512: two workgroups per CU, each a contiguous range of output rows (the tests lower it to force multi-row walks)
    if (n_items < 1) { *per_out = 0; return 0; }
    if (wgs < 1) wgs = 1;
    if (wgs > n_items) wgs = n_items;
    const int per = (n_items + wgs - 1) / wgs;
    *per_out = per;
    return (n_items + per - 1) / per;
}

static int stem_tpw(int hw) {                  // tiles of 32 pixels per wave: the largest count <= 8 with hw % (128 * tpw) == 0
    for (int t = 8; t >= 1; --t) if (hw % (128 * t) == 0) return t;
    return 0;
}

extern "C" int maed_stem7x7s2_supported(int H, int W) {
    if (H <= 0 || W <= 0 || (H & 1) || (W & 1)) return 0;
    const int Ho = H / 2, Wo = W / 2;
    return stem_tpw(Ho * Wo) > 0 && Wo % 16 == 0;
}

#define STEM_CHECK_GEOM(name) \
    MAED_CHECK_ARG(F > 0 && maed_stem7x7s2_supported(H, W), MAED_ERR_SHAPE, name ": needs even H, W with (H/2)*(W/2) %% 128 == 0 and (W/2) %% 16 == 0 (H=%d W=%d)", H, W); \
    const int Hp = H + 5, Wp = W + 6, Ho = H / 2, Wo = W / 2; \
    MAED_CHECK_ARG((int64_t)F * Hp * Wp * 8 < (1ll << 31) && (int64_t)F * Ho * Wo * 128 < (1ll << 32), MAED_ERR_SHAPE, name ": clip too large for 32-bit offsets (F=%d)", F)

// y (F, H/2, W/2, 64) = conv7x7 stride 2 of xp (F, H+5, W+6, 4) [maed_stem_input with c_stride 4] with w (64, 7, 7, 3) [channels_last standardised weight];
// wimg: 28 KB scratch for the fragment-major weight image; gn_sums (optional): (F, 32, 2) fp64 statistics of the GroupNorm behind, accumulated
extern "C" int maed_stem7x7s2_fwd(const void* xp, const void* w, void* wimg, void* y, double* gn_sums, int F, int H, int W, int dtype, void* stream) {
    MAED_CHECK_ARG(xp && w && wimg && y, MAED_ERR_ARG, "stem7x7s2_fwd: null pointer");
    MAED_CHECK_ARG(dtype == MAED_BF16, MAED_ERR_UNSUPPORTED, "stem7x7s2_fwd: bf16 only (dtype=%d)", dtype);
    STEM_CHECK_GEOM("stem7x7s2_fwd");
    MAED_CHECK_ARG(is_aligned(xp, 16) && is_aligned(wimg, 16) && is_aligned(y, 16), MAED_ERR_ALIGN, "stem7x7s2_fwd: 16-B alignment");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(stem_wimg_kernel, dim3(STEM_WIMG / 256), dim3(256), 0, s, (const bf16*)w, (bf16*)wimg);
    const int tpw = stem_tpw(Ho * Wo);
    const unsigned grid = (unsigned)((int64_t)F * Ho * Wo / (128 * tpw));
    if (gn_sums) hipLaunchKernelGGL(stem7x7s2_fwd_kernel<true>, dim3(grid), dim3(256), 0, s, (const bf16*)xp, (const bf16*)wimg, (bf16*)y, gn_sums, Hp, Wp, Ho, Wo, tpw);
    else hipLaunchKernelGGL(stem7x7s2_fwd_kernel<false>, dim3(grid), dim3(256), 0, s, (const bf16*)xp, (const bf16*)wimg, (bf16*)y, gn_sums, Hp, Wp, Ho, Wo, tpw);
    MAED_CHECK_LAUNCH("stem7x7s2_fwd");
    return MAED_OK;
}

// fp32 elements of scratch maed_stem7x7s2_wgrad wants for its per-workgroup partial results (0: geometry not covered)
extern "C" int maed_stem7x7s2_wgrad_scratch_floats(int F, int H, int W) {
    if (F <= 0 || !maed_stem7x7s2_supported(H, W)) return 0;
    int per = 0;
    return stem_wgrad_wgs(F * (H / 2), &per) * 64 * 147;
}

// dW (64, 7, 7, 3) fp32 += weight gradient from dy (F, H/2, W/2, 64) and xp (F, H+5, W+6, 4), channels_last bf16; scratch (optional): see above -- NULL: atomics
extern "C" int maed_stem7x7s2_wgrad(const void* dy, const void* xp, float* dW, void* scratch, int F, int H, int W, int dtype, void* stream) {
    MAED_CHECK_ARG(dy && xp && dW, MAED_ERR_ARG, "stem7x7s2_wgrad: null pointer");
    MAED_CHECK_ARG(dtype == MAED_BF16, MAED_ERR_UNSUPPORTED, "stem7x7s2_wgrad: bf16 only (dtype=%d)", dtype);
    STEM_CHECK_GEOM("stem7x7s2_wgrad");
    MAED_CHECK_ARG(is_aligned(dy, 16) && is_aligned(xp, 16) && is_aligned(scratch, 16), MAED_ERR_ALIGN, "stem7x7s2_wgrad: 16-B alignment");
    const int dy_rounds = (Wo * 8 + STEM_WG_THREADS - 1) / STEM_WG_THREADS, x_rounds = (7 * (Wp / 2) + STEM_WG_THREADS - 1) / STEM_WG_THREADS;
    const size_t lds = (size_t)2 * (dy_rounds + x_rounds) * STEM_WG_THREADS * 16;
    MAED_CHECK_ARG(lds <= 160 * 1024, MAED_ERR_SHAPE, "stem7x7s2_wgrad: image too wide for the LDS row buffers (W=%d)", W);
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)stem7x7s2_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
    const int n_items = F * Ho;
    int per = 0;
    const int wgs = stem_wgrad_wgs(n_items, &per);
    hipLaunchKernelGGL(stem7x7s2_wgrad_kernel, dim3(wgs), dim3(STEM_WG_THREADS), lds, (hipStream_t)stream, (const bf16*)dy, (const bf16*)xp, dW, (float*)scratch, Hp, Wp,
                       Ho, Wo, n_items, per);
    if (scratch) maed_wgrad_slots_reduce((const float*)scratch, dW, wgs, 64 * 147, (hipStream_t)stream);
    MAED_CHECK_LAUNCH("stem7x7s2_wgrad");
    return MAED_OK;
}
