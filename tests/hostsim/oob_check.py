"""TEST INFRASTRUCTURE: run simulator tests under AddressSanitizer and report out-of-bounds accesses inside kernel code.

On the GPU a kernel that reads or writes past the end of a tensor usually goes unnoticed: the address belongs to a neighbouring allocation of the
caching allocator, parity tests of the op itself stay green, and some other tensor is silently corrupted.  In the simulator global memory is ordinary
heap memory, so the same access is a heap-buffer-overflow with the kernel's source line.

    python tests/hostsim/oob_check.py [pytest args ...]         (default: the backbone / GEMM / attention simulator tests)
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DEFAULT = ["tests/test_hostsim_backbone.py", "tests/test_hostsim_gemm.py", "tests/test_hostsim_resnet.py", "tests/test_hostsim_attention.py", "tests/test_hostsim_elementwise.py"]


def asan_runtime():
    clang = os.environ.get("MAED_HOST_CXX", "/opt/rocm/lib/llvm/bin/clang++")
    out = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if os.path.isabs(out) and os.path.exists(out):
        return out
    hits = glob.glob(os.path.join(os.path.dirname(os.path.dirname(clang)), "lib", "clang", "*", "lib", "linux", "libclang_rt.asan-x86_64.so"))
    if not hits:
        sys.exit("no AddressSanitizer runtime next to " + clang)
    return hits[0]


PROBE = """
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from _hostsim import patched
from maed_amd import ops, _lib as L
A = torch.randn(130, 64).bfloat16(); B = torch.randn(64, 64).bfloat16()
out = torch.empty(%%d, 64, dtype=torch.bfloat16)
with patched():
    ops.gemm_nt(A, B, L.EPI_STORE, out=out)
""" % (ROOT, os.path.join(ROOT, "tests"))


def reports_in(log):
    reps = []
    for f in glob.glob(log + "*"):
        for rep in re.split(r"={60,}\n", open(f, errors="replace").read()):
            # a kernel frame shows up by source path when the report is symbolised, by module name when it is not
            if "ERROR: AddressSanitizer" in rep and ("maed_amd/csrc/" in rep or "libmaed_hostsim.so" in rep):
                reps.append(rep)
    return reps


def main():
    subprocess.run([sys.executable, os.path.join(HERE, "build_sim.py")], env=dict(os.environ, MAED_SIM_ASAN="1"), check=True, stdout=subprocess.DEVNULL)
    with tempfile.TemporaryDirectory() as td:
        def env_for(log):
            return dict(os.environ, MAED_SIM_ASAN="1", LD_PRELOAD=asan_runtime(),
                        ASAN_OPTIONS=f"detect_leaks=0:halt_on_error=0:log_path={log}:allocator_may_return_null=1:detect_odr_violation=0")
        # the checker proves itself first: a GEMM whose output tensor is one row short must be reported, the correctly sized one must not
        for rows, want in ((130, 0), (129, 1)):
            log = os.path.join(td, f"probe{rows}")
            subprocess.run([sys.executable, "-c", PROBE % rows], cwd=ROOT, env=env_for(log), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            got = len(reports_in(log))
            if (got > 0) != (want > 0):
                sys.exit(f"out-of-bounds checker self-test failed: output with {rows} rows of 130 gave {got} reports")
        print("self-test ok: a store past a short output tensor is reported, the full-size call is silent", flush=True)
        log = os.path.join(td, "asan")
        env = env_for(log)
        rc = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider"] + (sys.argv[1:] or DEFAULT), cwd=ROOT, env=env).returncode
        reports = reports_in(log)
        seen = set()
        for r in reports:
            frames = [l.strip() for l in r.splitlines() if "libmaed_hostsim.so" in l or ".hip:" in l or ".cuh:" in l][:8]
            key = tuple(frames[:2])
            if key in seen:
                continue
            seen.add(key)
            print("=" * 18 + "\n" + "\n".join(r.splitlines()[:3] + frames))
        print(f"pytest exit code {rc}; out-of-bounds reports inside kernel code: {len(reports)} ({len(seen)} distinct)")
        sys.exit(1 if reports else 0)


if __name__ == "__main__":
    main()
