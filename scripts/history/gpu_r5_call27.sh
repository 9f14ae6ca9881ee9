#!/bin/bash
# what clock do the GEMM kernels run at?  GRBM_GUI_ACTIVE (shader-engine busy cycles) per launch / launch duration = the effective clock; + MFMA busy cycles.
# separate passes per counter group, --kernel-trace only
set -u
cd "$(dirname "$0")/.."
ROOT=$(pwd); O=gpurun_out/r5c27; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for c in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $c | tr ' ' '+'); rm -rf /tmp/pmc_$tag
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python "$ROOT/scripts/x3p_micro.py" 2 1 > "$ROOT/$O/$tag.log" 2>&1)
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1); k=$(find /tmp/pmc_$tag -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/$tag.csv; [ -n "$k" ] && cp "$k" $O/${tag}_trace.csv
done
python - <<'PY'
import csv, glob, os, collections
O = "gpurun_out/r5c27"
for fn in sorted(glob.glob(O + "/*.csv")):
    if fn.endswith("_trace.csv"):
        continue
    tr = fn[:-4] + "_trace.csv"
    dur = {}
    if os.path.exists(tr):
        for r in csv.DictReader(open(tr)):
            dur[r.get("Dispatch_Id") or r.get("Correlation_Id")] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "gemm_nt_x3" not in k:
            continue
        key = (k[:90], r.get("Grid_Size"))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        d = dur.get(r.get("Dispatch_Id"))
        if d:
            agg[key]["us"].append(d)
    print("==", os.path.basename(fn))
    for key, cs in sorted(agg.items()):
        line = f"{key[0]:92s} grid {key[1]:>8s} "
        for c, v in cs.items():
            line += f" {c} {sum(v) / len(v):.4g}"
        print(line)
PY
