// What does the matrix pipe sustain?  Back-to-back v_mfma_f32_32x32x16_bf16 on registers only (no LDS, no memory), 4 independent accumulators per wave,
// 1 / 2 / 4 waves per SIMD on every CU, long enough (~1 ms, ~10 ms) for the clock to settle.  Prints TFLOP/s against the 2.5 PFLOP/s figure of 2.4 GHz x 256 CUs.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/peak/mfma_peak scripts/peak/mfma_peak.hip   (build container)   ->   run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float seed, const bf16x8_t* rnd) {
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 1e-3f); b[i] = (__bf16)(seed - i * 1e-3f); }
    if (rnd) { a = rnd[threadIdx.x]; b = rnd[256 + threadIdx.x]; }        // random operand bits: what the multipliers toggle on real data
    f32x16_t c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 12345.678f) out[0] = s;
}
int main() {
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs, clockRate %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    unsigned short h[512 * 8];
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(((x >> 16) & 0x807f) | (0x3f00 + ((x >> 9) & 0x0080))); }     // bf16 in +-[0.5, 2)
    bf16x8_t* rnd; hipMalloc(&rnd, sizeof(h)); hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
    for (const bf16x8_t* r : {(const bf16x8_t*)nullptr, (const bf16x8_t*)rnd}) for (int wps : {1, 2, 4}) for (int iters : {20000, 100000}) {
        const int grid = p.multiProcessorCount * wps;      // 256-thread workgroups = 4 waves = one per SIMD
        mfma_loop<<<grid, 256>>>(out, 100, 1.f, r); hipDeviceSynchronize();
        hipEventRecord(e0); mfma_loop<<<grid, 256>>>(out, iters, 1.f, r); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)grid * 4 * iters * 4 * 32768.0;
        printf("%s operands, waves per SIMD %d, %6d x 4 MFMAs per wave: %8.3f ms  %7.0f TFLOP/s  (%.0f cycles of a 2.4 GHz clock per MFMA and SIMD)\n", r ? "random  " : "constant", wps, iters, ms, flop / ms / 1e9,
               ms * 1e-3 * 2.4e9 / ((double)wps * iters * 4));
    }
    return 0;
}
