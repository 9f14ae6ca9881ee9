"""TEST INFRASTRUCTURE: run simulator tests under ThreadSanitizer and report data races inside kernel code.

GPU threads are host threads in the simulator and LDS / global memory are ordinary memory, so a kernel that reads LDS another
thread wrote without a barrier in between (or two threads storing to one address) is a host data race.  This script
  1. builds the TSan flavour of the simulator (MAED_SIM_TSAN=1, tests/hostsim/_build_tsan),
  2. proves the checker works: a deliberately racy kernel must be reported, its barrier-ed twin must not,
  3. runs the given pytest selection (default: the kernels written without GPU access) under TSan,
  4. prints every report that has a frame inside libmaed_hostsim.so (reports between uninstrumented libraries -- torch, OpenMP --
     are noise and are dropped) and exits non-zero if there is one.

    python tests/hostsim/race_check.py [pytest args ...]
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DEFAULT = [["tests/test_hostsim_attention.py", "-k", "long or coupling"],
           ["tests/test_hostsim_tail.py::test_lane_parallel_chain_kernels_are_bit_identical_to_the_serial_ones", "tests/test_hostsim_eval.py"]]


def tsan_runtime():
    clang = os.environ.get("MAED_HOST_CXX", "/opt/rocm/lib/llvm/bin/clang++")
    out = subprocess.run([clang, "-print-file-name=libclang_rt.tsan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if os.path.isabs(out) and os.path.exists(out):
        return out
    hits = glob.glob(os.path.join(os.path.dirname(os.path.dirname(clang)), "lib", "clang", "*", "lib", "linux", "libclang_rt.tsan-x86_64.so"))
    if not hits:
        sys.exit("no ThreadSanitizer runtime next to " + clang)
    return hits[0]


def run(cmd, log_prefix):
    env = dict(os.environ, MAED_SIM_TSAN="1", LD_PRELOAD=tsan_runtime(),
               TSAN_OPTIONS=f"report_signal_unsafe=0 halt_on_error=0 exitcode=0 history_size=4 log_path={log_prefix}")
    rc = subprocess.run(cmd, cwd=ROOT, env=env).returncode
    reports = []
    for f in glob.glob(log_prefix + "*"):
        for rep in re.split(r"={18}\n", open(f, errors="replace").read()):
            # a kernel race is between two simulator threads; reports that involve torch's own (OpenMP) worker threads are stack-address
            # reuse seen through uninstrumented synchronisation
            if "WARNING: ThreadSanitizer" in rep and "libmaed_hostsim.so" in rep and "libgomp" not in rep and "libtorch" not in rep:
                reports.append(rep)
    return rc, reports


def main():
    subprocess.run([sys.executable, os.path.join(HERE, "build_sim.py")], env=dict(os.environ, MAED_SIM_TSAN="1"), check=True, stdout=subprocess.DEVNULL)
    with tempfile.TemporaryDirectory() as td:
        probe = ("import ctypes, sys; sys.path.insert(0, %r); import build_sim; h = ctypes.CDLL(build_sim.build()); "
                 "buf = (ctypes.c_int * 64)(); h.hostsim_race_selftest(buf, %%d)" % HERE)
        _, clean = run([sys.executable, "-c", probe % 1], os.path.join(td, "ok"))
        _, racy = run([sys.executable, "-c", probe % 0], os.path.join(td, "racy"))
        if clean or not racy:
            sys.exit(f"race checker self-test failed: barrier-ed kernel {len(clean)} reports (want 0), racy kernel {len(racy)} reports (want > 0)")
        print(f"self-test ok: racy kernel reported ({len(racy)}), barrier-ed kernel silent", flush=True)
        rc, reports = 0, []
        for i, sel in enumerate([sys.argv[1:]] if sys.argv[1:] else DEFAULT):
            r, rep = run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider"] + sel, os.path.join(td, f"run{i}"))
            rc, reports = rc or r, reports + rep
        for r in reports[:10]:
            print("=" * 18 + "\n" + r)
        print(f"pytest exit code {rc}; data-race reports inside kernel code: {len(reports)}")
        sys.exit(1 if (rc or reports) else 0)


if __name__ == "__main__":
    main()
