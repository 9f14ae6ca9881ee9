"""kernel trace of vendor_kernels.py -> one line per shape: kernel name, workgroup / grid size, LDS, registers, median us"""
import csv
import statistics
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from vendor_kernels import SHAPES  # noqa: E402

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
cur, seen = None, {}
for r in rows:
    nm = r["Kernel_Name"]
    gs = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
    if "FillFunctor" in nm and gs % 1 == 0:
        # marker: the fill of 4096 * (i + 1) floats (vectorised by 4, 256 threads: grid = elements / 4 rounded to the block) -- recover i from the element count
        for i in range(len(SHAPES)):
            n = 4096 * (i + 1)
            if gs == n // 4:
                cur = i
        continue
    if cur is None or "Cijk" not in nm and "gemm" not in nm.lower():
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    seen.setdefault(cur, {}).setdefault(nm, {"us": [], "wg": r.get("Workgroup_Size") or r.get("Workgroup_Size_X"), "grid": gs, "lds": r.get("LDS_Block_Size"),
                                              "vgpr": r.get("VGPR_Count"), "agpr": r.get("Accum_VGPR_Count"), "sgpr": r.get("SGPR_Count"), "scratch": r.get("Scratch_Size")})["us"].append(d)
for i, (name, m, n, k, kind) in enumerate(SHAPES):
    fl = 2.0 * m * n * k / 1e6
    for nm, v in seen.get(i, {}).items():
        us = statistics.median(v["us"])
        print(f"{name:16s} {m:6d}x{n:5d}x{k:5d}  {us:8.1f} us {fl / us:7.1f} TF  x{len(v['us'])}  wg {v['wg']} grid {v['grid']} (= {int(v['grid']) // max(1, int(v['wg'] or 1))} workgroups) "
              f"lds {v['lds']} vgpr {v['vgpr']} agpr {v['agpr']} sgpr {v['sgpr']}\n      {nm}")
