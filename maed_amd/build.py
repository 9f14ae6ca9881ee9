"""Build libmaed_hip.so (gfx950) in-tree with hipcc.  `python -m maed_amd.build [--force]`.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
maed_amd/libmaed_hip.so is git-ignored but travels to the GPU box with the repo snapshot."""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libmaed_hip.so")
SOURCES = ["layernorm.hip", "gemm.hip", "gemm256.hip", "gemm_tn.hip", "attn_spatial.hip", "attn_temporal.hip", "elementwise.hip", "block.hip", "smpl.hip", "backbone.hip", "tail_bwd.hip", "loss.hip", "comm.hip", "eval_metrics.hip", "attn_long.hip", "gemm_x3.hip", "options.hip", "attn_x3.hip", "stem.hip", "conv3x3_rows.hip", "gemm_tn2.hip", "gemm_x3p.hip", "gemm_sk.hip", "gemm_tn_sk.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]


def source_hash():
    """sha256 over the kernel sources + the C-ABI header: stamps measurement files (profiles/*_pmc/traffic.json) so that bench.py
    quotes a PMC traffic figure only for the build it was measured on"""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cuh", ".h")))
    for f in files + [os.path.join(os.path.dirname(HERE), "include", "maed_hip.h")]:
        with open(f if os.path.isabs(f) else os.path.join(CSRC, f), "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    # every header of csrc/ is a dependency of every object (a header-only edit must never leave stale objects in the library)
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cuh", ".h"))) + [os.path.join(os.path.dirname(HERE), "include", "maed_hip.h")]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        r = subprocess.run([hipcc, *FLAGS, "-c", s, "-o", o], capture_output=True, text=True)
        return s, r

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for s, r in ex.map(compile_one, jobs):
            if verbose and (r.stderr.strip() or r.returncode):
                print(r.stderr, file=sys.stderr)
            if r.returncode:
                raise RuntimeError(f"hipcc failed on {s}")
    objs = [os.path.join(OBJ, src.replace(".hip", ".o")) for src in SOURCES]
    if force or jobs or _stale(LIB, objs):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
        if r.returncode:
            print(r.stderr, file=sys.stderr)
            raise RuntimeError("link failed")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1024:.0f} KiB, {len(jobs)} recompiled)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
