// TEST INFRASTRUCTURE: thread-local state of the host simulator + the error plumbing block.hip owns in the real library.
#include <hip/hip_runtime.h>
#include <stdarg.h>
namespace hostsim {
thread_local Idx t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local BlockCtx* t_block = nullptr;
thread_local int t_tid = 0;
thread_local void* t_dyn_lds = nullptr;
}
static thread_local char g_err[512] = "";
void maed_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
extern "C" const char* maed_last_error(void) { return g_err; }
extern "C" int maed_version(void) { return -1; }   // negative: this is the host simulator, never the product library
