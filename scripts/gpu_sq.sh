#!/bin/bash
# SQ-level counters of the attention forward kernel (where do the wave cycles go?)
set -u
cd "$(dirname "$0")/.."
ROOT=$(pwd); mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
python scripts/attn_micro.py 50 | tee gpurun_out/pmc/attn_micro.txt
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCC|TCP|TA|GRBM)_[A-Z0-9_]+" | sort -u > gpurun_out/pmc/counters_available.txt; wc -l gpurun_out/pmc/counters_available.txt
run() { tag=$1; shift; rm -rf /tmp/pmc_$tag
  (cd /tmp && timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$tag -o attn -- python "$ROOT/scripts/attn_micro.py" 6 > "$ROOT/gpurun_out/pmc/$tag.log" 2>&1)
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "$tag: no output"; tail -3 "$ROOT/gpurun_out/pmc/$tag.log"; return; }
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "attn_sp_fwd_mfma" in r.get("Kernel_Name", "") or "attn_long_fwd_mfma" in r.get("Kernel_Name", ""):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"  {k:28s} mean {sum(v) / len(v):16.1f}  (n={len(v)})")
PY
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run sq3 SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_WAVES GRBM_GUI_ACTIVE
