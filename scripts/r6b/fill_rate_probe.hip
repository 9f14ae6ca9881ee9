// Probe (round 6, second session): how many bytes per clock does ONE CU move from L2 into LDS -- by LDS-DMA (global_load_lds_dwordx4), by plain loads into registers +
// ds_write_b128, and by both at once?  One 512-thread workgroup per CU, every workgroup streams its own 1 MB window (L2-resident after the first pass) `reps` times
// into a 64 KB LDS ring; nothing is computed.  Build: hipcc --offload-arch=gfx950 -O3 -o fill_rate_probe fill_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define DMA16(gptr_, lds_ptr_) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"((const char*)(gptr_)), "s"((uint32_t)(uintptr_t)(lds_ptr_)) : "memory")

// MODE 0: LDS-DMA only; 1: register staging only; 2: half the bytes each way
template <int MODE>
__global__ __launch_bounds__(512) void fill_kernel(const char* __restrict__ src, size_t window, int reps, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];          // 64 KB: 8 pieces of 8 KB (512 threads x 16 B)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (size_t)blockIdx.x * window;
    const int pieces = (int)(window / 8192);
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int r = 0; r < reps; ++r) {
        for (int p = 0; p < pieces; p += 8) {
            // eight 8 KB pieces in flight, then wait for all of them (a deep ring's steady state: the probe measures the path, not a schedule)
            uint4 reg[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const char* g = base + (size_t)(p + q) * 8192 + tid * 16;
                char* l = lds + q * 8192;
                const bool dma = MODE == 0 || (MODE == 2 && (q & 1));
                if (dma) DMA16(g, l + wave * 1024);
                else reg[q] = *reinterpret_cast<const uint4*>(g);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bool dma = MODE == 0 || (MODE == 2 && (q & 1));
                if (!dma) *reinterpret_cast<uint4*>(lds + q * 8192 + tid * 16) = reg[q];
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const uint4 v = *reinterpret_cast<const uint4*>(lds + ((tid * 16 + p * 64) & 65535 & ~15));
            acc.x ^= v.x; acc.y ^= v.y;
            __builtin_amdgcn_s_barrier();
        }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = 1;
}

int main() {
    const int ncu = 256;
    char* d; CK(hipMalloc(&d, (size_t)ncu << 20)); CK(hipMemset(d, 1, (size_t)ncu << 20));
    int* sink; CK(hipMalloc(&sink, 4));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const double clk = prop.clockRate * 1e3;
    CK(hipFuncSetAttribute((const void*)fill_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)fill_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)fill_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (size_t window : {(size_t)65536, (size_t)262144, (size_t)1 << 20}) for (int grid : {256, 64}) for (int mode = 0; mode < 2; ++mode) {
        const int reps = (int)((40u << 20) / window);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9;
        for (int it = 0; it < 4; ++it) {
            CK(hipEventRecord(e0, 0));
            if (mode == 0) fill_kernel<0><<<grid, 512, 65536>>>(d, window, reps, sink);
            if (mode == 1) fill_kernel<1><<<grid, 512, 65536>>>(d, window, reps, sink);
            if (mode == 2) fill_kernel<2><<<grid, 512, 65536>>>(d, window, reps, sink);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double bytes = (double)grid * window * reps;
        printf("window %4zu KB per workgroup  %3d workgroups (one per CU)  %-44s %8.1f us   %6.1f GB/s per CU   %5.1f B/clk per CU (at %.2f GHz)   %6.2f TB/s chip\n", window >> 10, grid,
               mode == 0 ? "LDS-DMA (global_load_lds_dwordx4)" : mode == 1 ? "global_load_dwordx4 -> VGPR -> ds_write_b128" : "half the pieces each way", best * 1e3,
               bytes / grid / (best * 1e-3) / 1e9, bytes / grid / (best * 1e-3) / clk, clk / 1e9, bytes / (best * 1e-3) / 1e12);
    }
    return 0;
}
