"""TEST INFRASTRUCTURE: compile the decoder-tail / loss kernel SOURCES of libmaed_hip for x86 against the
host-simulation shim (tests/hostsim/hip/hip_runtime.h) -> tests/hostsim/_build/libmaed_hostsim.so.

The simulator checks kernel arithmetic and the ctypes/autograd wiring on a box without a GPU.  It is
loaded only by tests (tests/test_hostsim_*.py monkeypatch maed_amd._lib); the product never sees it.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "maed_amd", "csrc")
# MAED_SIM_TSAN=1: ThreadSanitizer build (separate directory).  GPU threads are host threads and LDS is ordinary memory, so a missing
# __syncthreads() / wave barrier in a kernel is a data race TSan reports with both stacks.  Run the tests with
#   LD_PRELOAD=$(dirname $(which clang))/../lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so MAED_SIM_TSAN=1 python -m pytest ...
# (tests/hostsim/race_check.py does all of that).
TSAN = os.environ.get("MAED_SIM_TSAN", "0") not in ("", "0")
# MAED_SIM_ASAN=1: AddressSanitizer build.  Global memory is ordinary heap memory in the simulator (tensors of the framework's CPU allocator), so a kernel
# that loads or stores past the end of a tensor -- invisible on the GPU, where it lands in a neighbouring allocation -- is a heap-buffer-overflow
# report with the kernel's source line (tests/hostsim/oob_check.py).
ASAN = os.environ.get("MAED_SIM_ASAN", "0") not in ("", "0")
SAN = ["-fsanitize=thread"] if TSAN else ["-fsanitize=address"] if ASAN else []
OUT_DIR = os.path.join(HERE, "_build_tsan" if TSAN else "_build_asan" if ASAN else "_build")
OUT = os.path.join(OUT_DIR, "libmaed_hostsim.so")
SOURCES = ["smpl.hip", "tail_bwd.hip", "loss.hip", "elementwise.hip", "layernorm.hip", "backbone.hip", "gemm.hip", "gemm256.hip", "gemm_tn.hip", "attn_spatial.hip", "attn_temporal.hip", "block.hip", "eval_metrics.hip", "attn_long.hip", "gemm_x3.hip", "options.hip", "attn_x3.hip", "stem.hip", "conv3x3_rows.hip", "gemm_tn2.hip", "gemm_x3p.hip", "gemm_sk.hip", "gemm_tn_sk.hip",
           "comm.hip"]      # (the RCCL wrapper against tests/hostsim/rccl/rccl.h: streams / events are no-ops here, the NCCL entry points come from whatever library the test names)
CLANG = os.environ.get("MAED_HOST_CXX", "/opt/rocm/lib/llvm/bin/clang++")


def build(force=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))] + [os.path.join(HERE, "sim_support.cpp")]
    deps = srcs + [os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(HERE, "rccl", "rccl.h")] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cuh", ".h"))) + [
        os.path.join(ROOT, "include", "maed_hip.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    flags = [CLANG, "-std=c++20", "-O1", "-fPIC", "-pthread", "-I", HERE, "-Wno-unused-value"]
    if SAN:
        flags += SAN + ["-g", "-fno-omit-frame-pointer"]

    def compile_one(src):     # one object per source, in parallel (the whole library is ~14 translation units)
        obj = os.path.join(OUT_DIR, os.path.basename(src) + ".o")
        hdrs = deps[len(srcs):]
        if force or not os.path.exists(obj) or any(os.path.getmtime(obj) < os.path.getmtime(d) for d in [src] + hdrs):
            subprocess.run(flags + ["-c", "-x", "c++", src, "-o", obj], check=True)
        return obj

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, srcs))
    subprocess.run([CLANG, "-shared", "-pthread"] + (SAN + ["-shared-libsan"] if SAN else []) + ["-o", OUT] + objs, check=True)
    return OUT


def build_fakerccl(force=False):
    """libfakerccl.so (tests/hostsim/fakerccl.cpp): the NCCL subset csrc/comm.hip binds, over shared memory between processes of one host"""
    src, out = os.path.join(HERE, "fakerccl.cpp"), os.path.join(OUT_DIR, "libfakerccl.so")
    deps = [src, os.path.join(HERE, "rccl", "rccl.h")]
    if force or not os.path.exists(out) or any(os.path.getmtime(out) < os.path.getmtime(d) for d in deps):
        os.makedirs(OUT_DIR, exist_ok=True)
        subprocess.run([CLANG, "-std=c++20", "-O1", "-fPIC", "-shared", "-pthread", "-I", HERE, src, "-o", out, "-lrt"], check=True)
    return out


if __name__ == "__main__":
    print(build(force=True))
