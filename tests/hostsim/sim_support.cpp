// TEST INFRASTRUCTURE: thread-local state of the host simulator (error plumbing and maed_version come from block.hip).
#include <hip/hip_runtime.h>
namespace hostsim {
thread_local Idx t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local BlockCtx* t_block = nullptr;
thread_local int t_tid = 0;
thread_local unsigned t_wop = 0;
thread_local void* t_dyn_lds = nullptr;
}

// Self-test of the race checker (MAED_SIM_TSAN build): a kernel with and without the barrier between an LDS write and the reads of
// it.  tests/hostsim/race_check.py expects ThreadSanitizer to stay silent for the first and to report the second.
static void selftest_kernel(int* out, int with_barrier) {
    __shared__ int cell[64];
    cell[threadIdx.x] = (int)threadIdx.x * 3;
    if (with_barrier) __syncthreads();
    out[threadIdx.x] = cell[(threadIdx.x + 1) & 63];
}
extern "C" void hostsim_race_selftest(int* out, int with_barrier) {
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, nullptr, out, with_barrier);
}
