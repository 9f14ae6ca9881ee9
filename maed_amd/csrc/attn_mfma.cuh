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
// The same fragment -- row e_base + (lane & 31) of the TRANSPOSE, k slots key_base + {4 hi .. 4 hi + 3} and + 8 -- straight from a ROW-MAJOR image
// X[key][e] (row stride KLD) with two transposing reads: inside a 16-lane group, lane 4j + t reads X[key0 + j][e0 + 4t .. + 3] and receives column
// (lane & 15) of that 4 x 16 block.  No transposed copy of the tile has to be staged (2-byte LDS writes, 16 per 16-byte chunk).
__device__ __forceinline__ bf16x8_t lds_frag_tr_rm(const unsigned short* X, int key_base, int e_base, int lane) {
    const int i = lane & 15;
    const unsigned short* p = X + (key_base + 4 * (lane >> 5) + (i >> 2)) * KLD + e_base + (lane & 16) + 4 * (i & 3);
    union { bf16x8_t v; uint2 u[2]; } f;
    auto lo = MAED_DS_READ_TR16(p);
    auto hi = MAED_DS_READ_TR16(p + 8 * KLD);
    __builtin_memcpy(&f.u[0], &lo, 8);
    __builtin_memcpy(&f.u[1], &hi, 8);
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


// The same result tile (32 rows of this wave x 64 head channels, acc in the D layout above) written as FULL 128-byte lines: through a 4 KB LDS patch of the wave
// (16-byte chunk c of row r at slot c ^ (r & 7): conflict-free ds_write_b64 / ds_read_b128), then four 16-byte stores per lane, eight lanes per row.  store_rowT
// writes 8-byte pieces, 32 different lines per instruction -- partial-line writes are what bounded the stem's first forward (26 of 91 us) and cost the one-kernel
// attention backward experiment 20 of 119 us.
// The patch must not be in use by anybody else (callers put a block barrier behind the last tile); the wave's LDS operations execute in order.
// row_ptr(r): element 0 of the destination of tile row r (0 .. 31), or nullptr for a row that is not written.
template <typename RowPtr>
__device__ __forceinline__ void store_tile_lines_at(unsigned short* patch, RowPtr row_ptr, const f32x16_t (&acc)[2], int lane, int accumulate) {
    const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<uint2*>(patch + l31 * 64 + (((4 * et + g) ^ (l31 & 7)) << 3) + 4 * hi) =
                make_uint2(pack_bf2(acc[et][4 * g], acc[et][4 * g + 1]), pack_bf2(acc[et][4 * g + 2], acc[et][4 * g + 3]));
    MAED_WAVE_LDS_SYNC();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = r * 64 + lane, row = n >> 3, c = n & 7;
        bf16* const rp = row_ptr(row);
        if (rp) {
            bf16* dst = rp + c * 8;
            uint4 v = *reinterpret_cast<const uint4*>(patch + row * 64 + ((c ^ (row & 7)) << 3));
            if (accumulate) {
                float a[8], b[8];
                ld8(dst, a);
                ld8(reinterpret_cast<const bf16*>(&v), b);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] += b[j];
                st8(dst, a);
            } else *reinterpret_cast<uint4*>(dst) = v;
        }
    }
    MAED_WAVE_LDS_SYNC();
}
// rows at a constant stride: row0_ptr = element 0 of the tile's first row, rows >= n_valid are not written
__device__ __forceinline__ void store_tile_lines(unsigned short* patch, bf16* row0_ptr, int64_t row_stride, int n_valid, const f32x16_t (&acc)[2], int lane, int accumulate) {
    store_tile_lines_at(patch, [=](int row) -> bf16* { return row < n_valid ? row0_ptr + row * row_stride : nullptr; }, acc, lane, accumulate);
}
