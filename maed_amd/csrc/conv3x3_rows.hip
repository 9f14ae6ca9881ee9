// Weight gradient of the stride-1 3x3 SAME convolution (resnetv2.py:74-93; the conv2 of every stage-1 bottleneck) for 64 -> 64 channels, one IMAGE ROW at a time:
//   dW[co][tap][ci] += sum_x dy[f, y, x][co] * x[f, y + ky - 1, x + kx - 1][ci]
// The general TN kernel (gemm_tn.hip) treats this as a (pixels x 64)^T (pixels x 576) GEMM on 128 x 128 output tiles: half of every tile is empty at 64 output
// channels, and the dy tile is transposed through registers once per K tile (five times).  Here a work item is one image row (f, y): its dy row and the three
// input rows y - 1 .. y + 1 are copied into LDS unchanged by LDS-DMA -- the input rows into a ring of row slots, so a workgroup walking consecutive rows
// fetches ONE new input row per item; the copies run three items ahead (14 KB per item and workgroup: with one item in flight the launch was bound by the DMA
// round trip, 82 us) -- and all nine taps are contracted from there: wave = tap, operands through ds_read_b64_tr_b16 of the row-major images (the
// tap's column shift is a pointer offset, the image borders are zero pixels that frame every row slot; rows above / below the image: the wave skips the item).
// Same scheme as the stem's weight gradient (stem.hip).  Chunk c of pixel slot s is stored at c ^ 4 * ((s >> 1) & 1): pixel rows are 128 bytes apart and a
// transposing read covers four consecutive pixels.
#include "common.cuh"
#include "prof.h"

#define R3_ROWPX 66                       // pixel slots per ring row: zero pixel, up to 64 image pixels, zero pixel
#define R3_ROW_ELEMS (R3_ROWPX * 64)
#define R3_DY_ELEMS (64 * 64)             // dy buffer: 64 pixel slots (slots >= W stay zero)
#define R3_THREADS 576                    // nine waves: one per tap
#define R3_RING 8                         // input-row slots: rows g - 1 .. g + 4 are live at item g (three in use, three on their way); 8 for the mask
#define R3_NDY 4                          // dy row buffers: item g in use, g + 1 .. g + 3 on their way
#define R3_LDS_BYTES ((R3_RING * R3_ROW_ELEMS + R3_NDY * R3_DY_ELEMS) * 2)       // 100 KB: one workgroup per CU

__global__ __launch_bounds__(R3_THREADS, 1) void conv3x3_wgrad_rows64_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x, float* __restrict__ dW,
                                                                               float* __restrict__ partial, int H, int W, int n_rows, int rows_per_wg) {
    MAED_DYN_SHARED(unsigned short, smem);
    const int tid = threadIdx.x, lane = tid & 63, tap = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5, i16 = lane & 15;
    const int ky = tap / 3, kx = tap - 3 * ky;
    const int g0 = blockIdx.x * rows_per_wg;
    int g1 = g0 + rows_per_wg;
    if (g1 > n_rows) g1 = n_rows;
    if (g0 >= g1) return;
    for (int i = tid; i < R3_LDS_BYTES / 16; i += R3_THREADS) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    unsigned short* const ring = smem;
    unsigned short* const dyb = smem + R3_RING * R3_ROW_ELEMS;
    const int row_chunks = W * 8, n_instr = row_chunks >> 6;          // W % 8 == 0 (host-checked): whole wave-instructions of 64 chunks
    const int64_t row_bytes = (int64_t)W * 128;
    // position p of a row's chunks = (pixel p >> 3, LDS chunk p & 7); the global chunk that belongs there undoes the slot swizzle
#define R3_ISSUE_X(g_) if (tap < n_instr) { const int g__ = (g_); const int p = tap * 64 + lane, px = p >> 3, c = p & 7; \
        MAED_LDS_DMA16((const char*)x + g__ * row_bytes, (uint32_t)((px * 8 + (c ^ ((((px + 1) >> 1) & 1) << 2))) * 16), ring + (g__ & (R3_RING - 1)) * R3_ROW_ELEMS + 64 + tap * 512); }
#define R3_ISSUE_DY(g_, b_) if (tap < n_instr) { const int g__ = (g_); const int p = tap * 64 + lane, px = p >> 3, c = p & 7; \
        MAED_LDS_DMA16((const char*)dy + g__ * row_bytes, (uint32_t)((px * 8 + (c ^ (((px >> 1) & 1) << 2))) * 16), dyb + ((b_) & (R3_NDY - 1)) * R3_DY_ELEMS + tap * 512); }

    f32x16_t acc[2][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][0][r] = 0.f; acc[0][1][r] = 0.f; acc[1][0][r] = 0.f; acc[1][1][r] = 0.f; }
    // transposing reads: pixel row 4 hi + (i16 >> 2) (+ 8) of a 16-pixel k-step, channels (lane & 16) + 4 (i16 & 3) .. (+ 32 for the second channel tile)
    const int prow = 4 * hi + (i16 >> 2), pcol = (lane & 16) + 4 * (i16 & 3);
    const int swa = ((prow >> 1) & 1) << 2, swb = (((prow + kx) >> 1) & 1) << 2;      // bit 1 of the pixel slot: unchanged by + 8 and + 16 s
    const int a_off0 = prow * 64 + ((((pcol >> 3) ^ swa) << 3) | (pcol & 7)), a_off1 = prow * 64 + (((((pcol + 32) >> 3) ^ swa) << 3) | (pcol & 7));
    const int b_off0 = (prow + kx) * 64 + ((((pcol >> 3) ^ swb) << 3) | (pcol & 7)), b_off1 = (prow + kx) * 64 + (((((pcol + 32) >> 3) ^ swb) << 3) | (pcol & 7));
    const int ksteps = (W + 15) >> 4;

    // prefetch of item k: its dy row and input row k + 1 (rows k - 1, k came with the items before); the first item also brings rows g0 - 1, g0
#define R3_PREFETCH(k_) { const int k__ = (k_); if (k__ < g1) { R3_ISSUE_DY(k__, k__ - g0); if (k__ + 1 < n_rows) R3_ISSUE_X(k__ + 1); } }
    if (g0 > 0) R3_ISSUE_X(g0 - 1);
    R3_ISSUE_X(g0);
    R3_PREFETCH(g0);
    R3_PREFETCH(g0 + 1);
    R3_PREFETCH(g0 + 2);
    for (int g = g0; g < g1; ++g) {
        const int b = g - g0;
        // item g's copies are complete once at most the two younger items' (two DMAs per wave each, when they were issued in full) are outstanding
        if (g + 2 < g1 && g + 3 < n_rows) { MAED_WAIT_VMCNT(4); } else { MAED_WAIT_VMCNT0(); }
        __syncthreads();                 // row g's images have landed for every wave; nobody still reads item g - 1's dy buffer and oldest ring slot
        R3_PREFETCH(g + 3);
        const int y = g % H, yy = y + ky - 1;
        if (yy < 0 || yy >= H) continue;                                  // wave-uniform: this tap's input row lies outside the image
        const unsigned short* da = dyb + (b & (R3_NDY - 1)) * R3_DY_ELEMS;
        const unsigned short* xb = ring + ((g + ky - 1) & (R3_RING - 1)) * R3_ROW_ELEMS;
        for (int s = 0; s < ksteps; ++s) {
            union { bf16x8_t v; uint2 u[2]; } a0, a1, b0, b1;
#define R3_FRAG(dst_, base_, off_) { auto lo__ = MAED_DS_READ_TR16((base_) + s * 1024 + (off_)); auto hi__ = MAED_DS_READ_TR16((base_) + s * 1024 + 512 + (off_)); \
            __builtin_memcpy(&dst_.u[0], &lo__, 8); __builtin_memcpy(&dst_.u[1], &hi__, 8); }
            R3_FRAG(a0, da, a_off0) R3_FRAG(a1, da, a_off1) R3_FRAG(b0, xb, b_off0) R3_FRAG(b1, xb, b_off1)
#undef R3_FRAG
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.v, b0.v, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.v, b1.v, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, b0.v, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, b1.v, acc[1][1], 0, 0, 0);
        }
    }
#undef R3_PREFETCH
#undef R3_ISSUE_X
#undef R3_ISSUE_DY
    // D[co][ci]: column ci = l31 (+ 32 b), rows (r & 3) + 8 (r >> 2) + 4 hi (+ 32 a);  dW (64, 9, 64).
    // With `partial` the workgroup stores its 64 x 576 block with plain stores into its own slot (summed by wgrad_slots_reduce_kernel): 256 workgroups adding
    // 147 KB each with atomics onto the SAME 147 KB cost 39 of the launch's 80 us (fp32 atomics on a hot spot: ~1 TB/s, the same at workgroup scope).
    float* const base = partial ? partial + (size_t)blockIdx.x * (64 * 576) : dW;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bt = 0; bt < 2; ++bt) {
            float* d = base + tap * 64 + 32 * bt + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* e = d + (32 * a + (r & 3) + 8 * (r >> 2) + 4 * hi) * 576;
                if (partial) *e = acc[a][bt][r]; else atomicAdd(e, acc[a][bt][r]);
            }
        }
}

// dW[e] += sum over workgroup slots w of partial[w][e]: 36864 elements, one per thread and slot chunk (grid.y chunks of <= 32 slots: coalesced 1 KB rows)
__global__ __launch_bounds__(256) void wgrad_slots_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dW, int n_slots, int n_elems) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_elems) return;
    const int w0 = blockIdx.y * 32, w1 = min(n_slots, w0 + 32);
    float t0 = 0.f, t1 = 0.f;
    int w = w0;
    for (; w + 1 < w1; w += 2) { t0 += partial[(size_t)w * n_elems + e]; t1 += partial[(size_t)(w + 1) * n_elems + e]; }
    if (w < w1) t0 += partial[(size_t)w * n_elems + e];
    atomicAdd(dW + e, t0 + t1);
}

bool maed_conv3x3_wgrad_rows64_ok(int F, int H, int W, int Cin, int Cout) {
    // (F >= 1: an empty batch has no rows to split over workgroups -- callers get "not applicable" and take the general route, which returns early on M == 0)
    return maed_opt(MAED_OPT_CONV3X3_ROWS_WGS) > 0 && F >= 1 && Cin == 64 && Cout == 64 && W % 8 == 0 && W >= 8 && W <= 64 && H >= 1 && (int64_t)F * H * W * 128 < (1ll << 31);
}

void maed_wgrad_slots_reduce(const float* partial, float* dW, int n_slots, int n_elems, hipStream_t stream) {
    hipLaunchKernelGGL(wgrad_slots_reduce_kernel, dim3((n_elems + 255) / 256, (n_slots + 31) / 32), dim3(256), 0, stream, partial, dW, n_slots, n_elems);
}

// workgroups (= partial-sum slots) of a launch over n_rows image rows
static int rows64_wgs(int n_rows, int* per_out) {
    int wgs = maed_opt(MAED_OPT_CONV3X3_ROWS_WGS);   // default 256, one per CU: every workgroup ends with a 147 KB partial result
    if (n_rows < 1 || wgs < 1) { *per_out = 0; return 0; }
    if (wgs > n_rows) wgs = n_rows;
    const int per = (n_rows + wgs - 1) / wgs;
    *per_out = per;
    return (n_rows + per - 1) / per;
}

// dW (64, 3, 3, 64) fp32 += weight gradient from dy, x (F, H, W, 64) channels_last bf16.  scratch (optional): maed_conv3x3_wgrad_rows64_scratch_floats(...) floats for
// the per-workgroup partial results (plain stores + one reduction pass); without it the workgroups add into dW with atomics.
int maed_conv3x3_wgrad_rows64_launch(const void* dy, const void* x, float* dW, void* scratch, int F, int H, int W, hipStream_t stream) {
    const int n_rows = F * H;
    int per = 0;
    const int wgs = rows64_wgs(n_rows, &per);
    if (wgs == 0) return MAED_OK;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_rows64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
    hipLaunchKernelGGL(conv3x3_wgrad_rows64_kernel, dim3(wgs), dim3(R3_THREADS), R3_LDS_BYTES, stream, (const bf16*)dy, (const bf16*)x, dW, (float*)scratch, H, W, n_rows, per);
    if (scratch) maed_wgrad_slots_reduce((const float*)scratch, dW, wgs, 64 * 576, stream);
    MAED_CHECK_LAUNCH("conv3x3_wgrad(rows)");
    return MAED_OK;
}

extern "C" int maed_conv3x3_wgrad_rows64_scratch_floats(int F, int H, int W, int Cin, int Cout) {
    if (!maed_conv3x3_wgrad_rows64_ok(F, H, W, Cin, Cout)) return 0;
    int per = 0;
    return rows64_wgs(F * H, &per) * 64 * 576;
}

extern "C" int maed_conv3x3_wgrad_rows64(const void* dy, const void* x, float* dW, void* scratch, int F, int H, int W, int dtype, void* stream) {
    MAED_CHECK_ARG(dy && x && dW, MAED_ERR_ARG, "conv3x3_wgrad_rows64: null pointer");
    MAED_CHECK_ARG(dtype == MAED_BF16, MAED_ERR_UNSUPPORTED, "conv3x3_wgrad_rows64: bf16 only (dtype=%d)", dtype);
    MAED_CHECK_ARG(F > 0 && maed_conv3x3_wgrad_rows64_ok(F, H, W, 64, 64), MAED_ERR_SHAPE, "conv3x3_wgrad_rows64: needs W %% 8 == 0, 8 <= W <= 64 (F=%d H=%d W=%d)", F, H, W);
    MAED_CHECK_ARG(is_aligned(dy, 16) && is_aligned(x, 16) && is_aligned(scratch, 16), MAED_ERR_ALIGN, "conv3x3_wgrad_rows64: 16-B alignment");
    ProfScope prof__(PROF_TN_CONV, stream, 2.0 * (double)F * H * W * 64 * 576, 2.0 * (double)F * H * W * 128 + 36.0 * 64 * 64);
    return maed_conv3x3_wgrad_rows64_launch(dy, x, dW, scratch, F, H, W, (hipStream_t)stream);
}
