"""Static audit of every kernel in maed_amd/csrc: registers, AGPR use, scratch (spills), LDS, launch bound, and the occupancy
those imply -- from the compiler's own metadata (hipcc -S, no GPU needed).  Flags the things that cost time silently:
scratch inside a hot kernel, MFMA accumulators parked in AGPRs (v_accvgpr round trips), one wave per SIMD.

    python scripts/isa_audit.py [file.hip ...]        # default: all of maed_amd/csrc/*.hip
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "maed_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def short(name):
    """readable kernel name from the mangled one (no demangler in the image): drop the _Z prefix / nested-name lengths"""
    m = re.match(r"_ZN?(?:12_GLOBAL__N_1)?(\d+)", name)
    if not m:
        return name[:48]
    n = int(m.group(1))
    start = m.end()
    rest = name[start + n:]
    tmpl = re.match(r"I([A-Za-z0-9_]*?)E(?:E|v|P)", rest)
    return (name[start:start + n] + ("<" + tmpl.group(1) + ">" if tmpl and rest.startswith("I") else ""))[:60]


def audit(path):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", path, "-o", out],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    acc = {}
    cur = None
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
        elif cur and "v_accvgpr_" in line:
            acc[cur] = acc.get(cur, 0) + 1
    rows = []
    for blk in text[text.index("amdhsa.kernels:"):].split("  - .agpr_count:")[1:]:
        g = lambda key: re.search(r"\." + key + r":\s+(\S+)", blk).group(1)
        name = g("name")
        v, a = int(g("vgpr_count")), int(blk.split()[0])
        lds, mx, sc = int(g("group_segment_fixed_size")), int(g("max_flat_workgroup_size")), int(g("private_segment_fixed_size"))
        waves = max(1, min(8, 512 // max(v, 1)))                     # per SIMD, by the unified register file
        flags = []
        if sc:
            flags.append(f"SCRATCH {sc} B")
        if acc.get(name):
            flags.append(f"{acc[name]} v_accvgpr moves")
        if waves == 1:
            flags.append("1 wave/SIMD by registers")
        rows.append((os.path.basename(path), short(name), v, a, lds, mx, waves, "; ".join(flags)))
    return rows


if __name__ == "__main__":
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    print(f"{'file':18s} {'kernel':60s} {'vgpr':>5s} {'agpr':>5s} {'lds':>7s} {'maxwg':>6s} {'w/SIMD':>6s}  flags")
    for f in files:
        for r in audit(f):
            print(f"{r[0]:18s} {r[1]:60s} {r[2]:5d} {r[3]:5d} {r[4]:7d} {r[5]:6d} {r[6]:6d}  {r[7]}")
