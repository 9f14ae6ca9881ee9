// TEST INFRASTRUCTURE: thread-local state of the host simulator (error plumbing and maed_version come from block.hip).
#include <hip/hip_runtime.h>
namespace hostsim {
thread_local Idx t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local BlockCtx* t_block = nullptr;
thread_local int t_tid = 0;
thread_local void* t_dyn_lds = nullptr;
}
