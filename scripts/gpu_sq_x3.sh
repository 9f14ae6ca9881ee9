#!/bin/bash
# SQ counters of one split-bf16 kernel launch family:   gpu_sq_x3.sh [group=nt] [shape=sq4k] [mode=bf16x3] [kernel-name substring=gemm_nt_x3]
set -u
cd "$(dirname "$0")/.."
ROOT=$(pwd); mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
GROUP=${1:-nt}; SHAPE=${2:-sq4k}; MODE=${3:-bf16x3}; KNAME=${4:-gemm_nt_x3}
run() { tag=$1; shift; rm -rf /tmp/pmc_$tag
  (cd /tmp && timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$tag -o g -- python "$ROOT/scripts/x3_micro.py" 5 $GROUP $SHAPE $MODE > "$ROOT/gpurun_out/pmc/$tag.log" 2>&1)
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "$tag: no output"; tail -3 "$ROOT/gpurun_out/pmc/$tag.log"; return; }
  python - "$f" "$KNAME" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r.get("Kernel_Name", ""):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"  {k:28s} mean {sum(v) / len(v):16.1f}  (n={len(v)})")
PY
}
echo "== $GROUP $SHAPE $MODE ($KNAME)"
run g1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run g2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run g3 SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM
