// Fragment helpers shared by the MFMA attention kernels (attn_spatial.hip, attn_temporal.hip).
#pragma once
#include "common.cuh"

#define KLD 72  // row stride of row-major LDS images (elements): 144 B, conflict-free ds_read_b128

__device__ __forceinline__ bf16x8_t lds_frag_tr(const unsigned short* base) {  // 4 + 4 elements, 8 apart
    union { bf16x8_t v; uint2 u[2]; } f;
    f.u[0] = *reinterpret_cast<const uint2*>(base);
    f.u[1] = *reinterpret_cast<const uint2*>(base + 8);
    return f.v;
}
__device__ __forceinline__ bf16x8_t pack_frag(const f32x16_t& x, int st) {
    union { bf16x8_t v; uint32_t u[4]; } f;
#pragma unroll
    for (int j = 0; j < 4; ++j) f.u[j] = pack_bf2(x[8 * st + 2 * j], x[8 * st + 2 * j + 1]);
    return f.v;
}
__device__ __forceinline__ void store_rowT(bf16* row, const f32x16_t (&acc)[2], int hi, int accumulate) {
    // acc[et][4g+i] = value at e = et*32 + 8g + 4hi + i of this lane's row
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int e0 = et * 32 + 8 * g + 4 * hi;
            float v[4] = {acc[et][4 * g], acc[et][4 * g + 1], acc[et][4 * g + 2], acc[et][4 * g + 3]};
            if (accumulate) { float o[4]; ld4(row + e0, o); v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3]; }
            st4(row + e0, v);
        }
}

