"""kernel trace (+ optional counter csv) of attn_residency.py -> per (kernel kind, condition): median duration of the attention launches, FETCH_SIZE per launch"""
import csv
import statistics
import sys

trace, pmc = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None)
KIND = ["spatial fwd", "temporal fwd", "spatial bwd", "temporal bwd"]
COND = ["in situ (behind the qkv GEMM)", "cold (1 GB of stores in between)", "hot (second launch on the same tensor)"]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
fetch = {}
if pmc:
    for r in csv.DictReader(open(pmc)):
        if r.get("Counter_Name") == "FETCH_SIZE":
            fetch[r["Dispatch_Id"]] = float(r["Counter_Value"])
cur, dur, fs = None, {}, {}
for r in rows:
    nm, gs = r["Kernel_Name"], int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
    if "FillFunctor" in nm and gs % 1024 == 0 and gs // 1024 <= 16 and gs < 1 << 20:
        i = gs // 1024 - 1
        cur = None if i == 15 else i
        continue
    if cur is None or "attn" not in nm:
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    dur.setdefault(cur, {}).setdefault(nm.split("(")[0][:48], []).append(d)
    did = r.get("Dispatch_Id")
    if did in fetch:
        fs.setdefault(cur, {}).setdefault(nm.split("(")[0][:48], []).append(fetch[did])
print("# qkv (128 frames x 197 tokens x 1536, bf16) = 77.5 MB; o 25.8 MB; spatial fwd algorithmic bytes 104 MB; FETCH_SIZE in KB as reported (x2 = bytes of a wide stream)")
for kind in range(4):
    for cond in range(3):
        key = cond * 4 + kind
        for nm, v in dur.get(key, {}).items():
            f = fs.get(key, {}).get(nm)
            print(f"{KIND[kind]:13s} {COND[cond]:42s} {nm:48s} {statistics.median(v):8.1f} us (n={len(v)})" + (f"   FETCH_SIZE {statistics.median(f):10.0f} KB" if f else ""))
