// Probe (round 6, second session): what do closing fp32 atomics cost at agent scope vs at workgroup scope (performed in the issuing XCD's L2)
// when every workgroup that adds into an address sits on the same XCD?  Also reports the XCC_ID of every workgroup (is it blockIdx % 8?).
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o atomic_scope_probe atomic_scope_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void xcc_kernel(int* out) {
    if (threadIdx.x == 0) { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); out[blockIdx.x] = (int)(v & 0xf); }
}

// every workgroup adds a 128 x 128 fp32 tile (64 KB) into tile (blockIdx % tiles) of dst, like the weight-gradient kernel's epilogue:
// 4 waves, wave w owns a 64 x 64 quadrant as 2 x 2 accumulators of 32 x 32, lane = column
template <int SCOPE>
__global__ __launch_bounds__(256) void add_kernel(float* dst, int tiles, int ld, int spin) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hi = lane >> 5;
    const int tile = blockIdx.x % tiles;
    const int tpr = ld / 128;
    float* base = dst + (size_t)(tile / tpr) * 128 * ld + (tile % tpr) * 128;
    float v = 1.0f;
    for (int i = 0; i < spin; ++i) v = __builtin_fmaf(v, 1.0f, 0.0f);   // nothing: placeholder for a main loop
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wave >> 1) * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, col = (wave & 1) * 64 + b * 32 + l31;
                if (SCOPE == 0) atomicAdd(base + (size_t)row * ld + col, v);
                else if (SCOPE == 1) __hip_atomic_fetch_add(base + (size_t)row * ld + col, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if (SCOPE == 2) __hip_atomic_fetch_add(base + (size_t)row * ld + col, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                else base[(size_t)row * ld + col] = v;   // plain store: the floor
            }
}


// the alternative: every workgroup stores its 64 KB tile as a register image (write-through), publishes a flag, waits for the z - 1 other parts of its
// output tile, then sums ITS 1/z of the tile over all z images in fixed order and adds that to dst with plain loads / stores (each element has one owner).
__device__ __forceinline__ void st_wt(float* p, float4 v) { typedef float f32x4 __attribute__((ext_vector_type(4))); const f32x4 x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(x) : "memory"); }
__global__ __launch_bounds__(256) void xchg_kernel(float* dst, int tiles, int ld, int z, float* slabs, unsigned* flags, unsigned epoch, int* fail) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int tile = blockIdx.x / z, bz = blockIdx.x % z;
    const int tpr = ld / 128;
    float* base = dst + (size_t)(tile / tpr) * 128 * ld + (tile % tpr) * 128;
    float* slab = slabs + (size_t)blockIdx.x * 16384;
#pragma unroll
    for (int i = 0; i < 16; ++i) st_wt(slab + ((size_t)i * 256 + tid) * 4, make_float4(1.f, 1.f, 1.f, 1.f));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flags + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __shared__ int ok_s;
    if (tid == 0) ok_s = 1;
    __syncthreads();
    if (tid < 64) {
        bool ok = true;
        for (int p = tid; p < z; p += 64) {
            unsigned spins = 0;
            while (__hip_atomic_load(flags + tile * z + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) { __builtin_amdgcn_s_sleep(2); if (++spins > (1u << 22)) { ok = false; break; } }
        }
        if (!ok) { ok_s = 0; atomicAdd(fail, 1); }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // my slice: float4 indices [bz * 4096 / z, (bz + 1) * 4096 / z)
    const int per = 4096 / z;   // z divides 4096 here
    const float* tslab = slabs + (size_t)tile * z * 16384;
    for (int j = bz * per + tid; j < (bz + 1) * per; j += 256) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p0 = 0; p0 < z; p0 += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(tslab + (size_t)(p0 + u) * 16384 + (size_t)j * 4);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        const int i = j >> 8, t = j & 255, w = t >> 6, ln = t & 63, a = i >> 3, b = (i >> 2) & 1, q4 = i & 3;
        const int row = (w >> 1) * 64 + a * 32 + 8 * q4 + 4 * (ln >> 5), col = (w & 1) * 64 + b * 32 + (ln & 31);
        float* d = base + (size_t)row * ld + col;
        d[0] += s.x; d[ld] += s.y; d[2 * (size_t)ld] += s.z; d[3 * (size_t)ld] += s.w;
    }
}

int main() {
    int n = 2048;
    int* dx; CK(hipMalloc(&dx, n * 4));
    xcc_kernel<<<n, 64>>>(dx);
    std::vector<int> hx(n); CK(hipMemcpy(hx.data(), dx, n * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < n; ++i) bad += hx[i] != i % 8;
    printf("XCC_ID of workgroup b == b %% 8 for %d of %d workgroups; first 16:", n - bad, n);
    for (int i = 0; i < 16; ++i) printf(" %d", hx[i]);
    printf("\n");
    // two streams at once: does a concurrent dispatch change the mapping?
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    int* dx2; CK(hipMalloc(&dx2, n * 4));
    for (int rep = 0; rep < 4; ++rep) { xcc_kernel<<<n, 64, 0, s1>>>(dx); xcc_kernel<<<n, 64, 0, s2>>>(dx2); }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hx.data(), dx2, n * 4, hipMemcpyDeviceToHost));
    bad = 0; for (int i = 0; i < n; ++i) bad += hx[i] != i % 8;
    printf("with a concurrent dispatch on another stream: %d of %d as b %% 8\n", n - bad, n);

    const int ld = 1024;
    for (int tiles : {16, 48, 64}) {
        const int rows = tiles * 128 * 128 / ld;
        float* d; CK(hipMalloc(&d, (size_t)rows * ld * 4));
        for (int wgs : {256, 512, 1024}) {
            // tiles % 8 == 0: workgroup b -> tile b % tiles -> every adder of a tile has the same b % 8
            for (int scope = 0; scope < 4; ++scope) {
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                float best = 1e9;
                bool ok = true;
                for (int rep = 0; rep < 6; ++rep) {
                    CK(hipMemsetAsync(d, 0, (size_t)rows * ld * 4, 0));
                    CK(hipEventRecord(e0, 0));
                    if (scope == 0) add_kernel<0><<<wgs, 256>>>(d, tiles, ld, 0);
                    if (scope == 1) add_kernel<1><<<wgs, 256>>>(d, tiles, ld, 0);
                    if (scope == 2) add_kernel<2><<<wgs, 256>>>(d, tiles, ld, 0);
                    if (scope == 3) add_kernel<3><<<wgs, 256>>>(d, tiles, ld, 0);
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
                }
                if (scope < 3) {
                    std::vector<float> h((size_t)rows * ld); CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
                    const float want = (float)(wgs / tiles + ((wgs % tiles) ? 0 : 0));
                    size_t nbad = 0; for (size_t i = 0; i < h.size(); ++i) { const int tile = (int)(i / ld / 128) * (ld / 128) + (int)(i % ld) / 128; const float w = (float)(wgs / tiles + (tile < wgs % tiles ? 1 : 0)); nbad += h[i] != w; }
                    ok = nbad == 0; (void)want;
                    if (!ok) printf("   MISMATCH: %zu elements\n", nbad);
                }
                printf("tiles %3d  wgs %4d  %-22s %7.1f us  (%5.1f MB added)%s\n", tiles, wgs, scope == 0 ? "agent-scope atomicAdd" : scope == 1 ? "workgroup-scope" : scope == 2 ? "wavefront-scope" : "plain stores", best * 1000.f, wgs * 65536.0 / 1e6, ok ? "" : "  WRONG");
            }
        }
        CK(hipFree(d));
    }

    {   // exchange-reduce against atomics
        float* slabs; unsigned* flags; int* fail; CK(hipMalloc(&slabs, (size_t)1024 * 65536)); CK(hipMalloc(&flags, 4096)); CK(hipMemset(flags, 0, 4096)); CK(hipMalloc(&fail, 4)); CK(hipMemset(fail, 0, 4));
        unsigned epoch = 0;
        for (int cfg = 0; cfg < 5; ++cfg) {
            const int tiles = cfg == 0 ? 16 : cfg == 1 ? 48 : cfg == 2 ? 64 : cfg == 3 ? 4 : 16, z = cfg == 0 ? 16 : cfg == 1 ? 8 : cfg == 2 ? 8 : cfg == 3 ? 64 : 32;
            const int rows = tiles * 128 * 128 / ld;
            float* d; CK(hipMalloc(&d, (size_t)rows * ld * 4));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float best = 1e9;
            for (int rep = 0; rep < 8; ++rep) {
                CK(hipMemsetAsync(d, 0, (size_t)rows * ld * 4, 0));
                CK(hipEventRecord(e0, 0));
                xchg_kernel<<<tiles * z, 256>>>(d, tiles, ld, z, slabs, flags, ++epoch, fail);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            std::vector<float> h((size_t)rows * ld); CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
            size_t nbad = 0; for (size_t i = 0; i < h.size(); ++i) nbad += h[i] != (float)z;
            int hf; CK(hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost));
            printf("exchange-reduce  tiles %3d  z %3d  wgs %4d  %7.1f us  (%5.1f MB of parts)  wrong elements %zu  timeouts %d\n", tiles, z, tiles * z, best * 1000.f, tiles * z * 65536.0 / 1e6, nbad, hf);
            CK(hipFree(d));
        }
    }
    return 0;
}
