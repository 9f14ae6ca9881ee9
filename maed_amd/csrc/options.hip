// Process-wide options of libmaed_hip (maed_set_option / maed_get_option, include/maed_hip.h).  They replace the environment variables the
// library used to read at call time: the host (maed_amd/_lib.py) translates its own configuration into option values once after loading the
// library; the kernels' launchers read them through maed_opt().
#include "common.cuh"
#include "gemm_x3.h"
#include <atomic>
#include <string.h>

static std::atomic<int> g_opt[MAED_OPT_COUNT] = {
    {0},      // MAED_OPT_F32_MATMUL: exact
    {1},      // MAED_OPT_SIDE_STREAM
    {0},      // MAED_OPT_TN_TARGET_WGS: 0 = the built-in heuristic (gemm_tn.hip maed_tn_splits)
    {0},      // MAED_OPT_ABLATE
    {1},      // MAED_OPT_GN_BWD_ONEPASS
    {0},      // MAED_OPT_F32_BWD_X1
    {1},      // MAED_OPT_ST_FUSED
    {256},    // MAED_OPT_CONV3X3_ROWS_WGS
    {512},    // MAED_OPT_STEM_WGRAD_WGS
    {0},      // MAED_OPT_LBS_FRAMES: auto
    {1},      // MAED_OPT_TN_DMA
    {6},      // MAED_OPT_X3_PLANES: 256 x 256 tiles
    {0},      // MAED_OPT_X3_PLANES_LN: measured neutral in the train step (profiles/r05_x3p_micro.txt)
    {1},      // MAED_OPT_SK: heuristic
    {0},      // MAED_OPT_SK_GRID: one workgroup per CU
    {1},      // MAED_OPT_TN_SK
    {0},      // MAED_OPT_CONV3X3_NARROW_WGS
    {1},      // MAED_OPT_CONV3X3_FRAME
};

extern "C" int maed_init(int device) {
#ifndef MAED_HOSTSIM
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { (void)hipGetLastError(); maed_set_error("init: no HIP device %d", device); return MAED_ERR_ARG; }
    if (!strstr(prop.gcnArchName, "gfx950")) {
        maed_set_error("init: device %d is %s; libmaed_hip is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return MAED_ERR_UNSUPPORTED;
    }
#else
    (void)device;
#endif
    (void)maed_fault_word();
#ifndef MAED_HOSTSIM
    if (maed_init_runtime() != MAED_OK) { maed_set_error("init: could not create the library's side streams / event rings"); return MAED_ERR_LAUNCH; }
#endif
    return MAED_OK;
}

// ---- device-fault word (common.cuh maed_report_fault): one uint32 in pinned, device-mapped host memory -- the kernels add to it with system scope, the host reads
// it without a synchronisation.  Allocated by maed_init (or on first use); never freed.
static std::atomic<uint32_t*> g_fault{nullptr};
static std::atomic<int> g_fault_told{0};
uint32_t* maed_fault_word(void) {
    uint32_t* w = g_fault.load(std::memory_order_acquire);
    if (w) return w;
#ifdef MAED_HOSTSIM
    uint32_t* n = (uint32_t*)calloc(16, sizeof(uint32_t));
#else
    uint32_t* n = nullptr;
    if (hipHostMalloc((void**)&n, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess || !n) { (void)hipGetLastError(); return nullptr; }
    memset(n, 0, 64);
#endif
    uint32_t* expected = nullptr;
    if (!g_fault.compare_exchange_strong(expected, n, std::memory_order_acq_rel)) {
#ifdef MAED_HOSTSIM
        free(n);
#else
        (void)hipHostFree(n);
#endif
        return expected;
    }
    return n;
}
bool maed_fault_seen(const char* who) {
    uint32_t* w = g_fault.load(std::memory_order_acquire);
    if (!w || *(volatile uint32_t*)w == 0) return false;
    if (!g_fault_told.exchange(1))
        maed_set_error("%s: %u frame-barrier timeout(s) on this device (workgroups of a frame were not co-resident: shared GPU / preemption); the affected results were "
                       "NaN-poisoned, the one-pass GroupNorm backward and the fused attentive addition now run as their multi-launch forms", who, *(volatile uint32_t*)w);
    return true;
}
// (maed_last_error is thread-local and the call that noticed the fault may have run on another thread -- autograd's backward worker: a non-zero answer also leaves the
//  explanation as THIS thread's last error)
extern "C" int maed_device_faults(void) {
    uint32_t* w = g_fault.load(std::memory_order_acquire);
    const int n = w ? (int)*(volatile uint32_t*)w : 0;
    if (n > 0)
        maed_set_error("%d frame-barrier timeout(s) on this device (workgroups of a frame were not co-resident: shared GPU / preemption); the affected results were "
                       "NaN-poisoned, the one-pass GroupNorm backward and the fused attentive addition run as their multi-launch forms from the next call on", n);
    return n;
}
extern "C" int maed_device_faults_clear(void) { uint32_t* w = g_fault.load(std::memory_order_acquire); if (w) *(volatile uint32_t*)w = 0; g_fault_told.store(0); return MAED_OK; }

extern "C" int maed_set_option(int key, int value) {
    MAED_CHECK_ARG(key >= 0 && key < MAED_OPT_COUNT, MAED_ERR_ARG, "set_option: unknown option %d", key);
    if (key == MAED_OPT_F32_MATMUL) MAED_CHECK_ARG(value >= 0 && value <= 2, MAED_ERR_ARG, "set_option: MAED_OPT_F32_MATMUL takes 0 (exact), 1 (bf16x3) or 2 (bf16x6)");
    if (key == MAED_OPT_TN_TARGET_WGS) MAED_CHECK_ARG(value == 0 || value >= 64, MAED_ERR_ARG, "set_option: MAED_OPT_TN_TARGET_WGS must be 0 (heuristic) or >= 64");
    if (key == MAED_OPT_CONV3X3_ROWS_WGS || key == MAED_OPT_STEM_WGRAD_WGS) MAED_CHECK_ARG(value >= 0 && value <= 65535, MAED_ERR_ARG, "set_option: workgroup count out of range");
    if (key == MAED_OPT_LBS_FRAMES) MAED_CHECK_ARG(value == 0 || value == 4 || value == 8 || value == 16, MAED_ERR_ARG, "set_option: MAED_OPT_LBS_FRAMES takes 0 (auto), 4, 8 or 16");
    if (key == MAED_OPT_X3_PLANES) MAED_CHECK_ARG(value == 0 || (value >= 2 && value <= 7 && value != 3), MAED_ERR_ARG, "set_option: MAED_OPT_X3_PLANES takes 0 (fp32 operands) or a kernel variant 2, 4, 5, 6, 7");
    if (key == MAED_OPT_SK) MAED_CHECK_ARG(value >= 0 && value <= 3, MAED_ERR_ARG, "set_option: MAED_OPT_SK takes 0 (off), 1 (heuristic), 2 (no K cuts) or 3 (always)");
    if (key == MAED_OPT_TN_SK) MAED_CHECK_ARG(value == 0 || value == 1, MAED_ERR_ARG, "set_option: MAED_OPT_TN_SK takes 0 or 1");
    if (key == MAED_OPT_SK_GRID) MAED_CHECK_ARG(value >= 0 && value <= 512, MAED_ERR_ARG, "set_option: MAED_OPT_SK_GRID takes 0 (one workgroup per CU) or a workgroup count <= 512");
    g_opt[key].store(value, std::memory_order_relaxed);
    return MAED_OK;
}

extern "C" int maed_get_option(int key) {
    if (key < 0 || key >= MAED_OPT_COUNT) { maed_set_error("get_option: unknown option %d", key); return MAED_ERR_ARG; }
    return g_opt[key].load(std::memory_order_relaxed);
}

int maed_opt(int key) { return g_opt[key].load(std::memory_order_relaxed); }

int maed_x3_planes(void) {
    const int v = maed_opt(MAED_OPT_F32_MATMUL);
    return v == 1 ? 2 : v == 2 ? 3 : 0;
}
