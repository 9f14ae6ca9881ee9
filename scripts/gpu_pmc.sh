#!/bin/bash
# HBM traffic of the attention kernel from PMC counters: separate passes for FETCH_SIZE and WRITE_SIZE
# (MI355X_MICROARCH.md: FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2 -> not both in one pass; kernel-trace only).
set -u
cd "$(dirname "$0")/.."
ROOT=$(pwd); mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
python scripts/attn_micro.py 50 | tee gpurun_out/pmc/attn_micro.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o attn -- python "$ROOT/scripts/attn_micro.py" 8 > "$ROOT/gpurun_out/pmc/$c.log" 2>&1)
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  echo "== $c: $f"; [ -n "$f" ] && head -3 "$f" && python - "$f" "$c" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "attn_sp_fwd_mfma" in r.get("Kernel_Name", "")]
vals = [float(r["Counter_Value"]) for r in rows if r.get("Counter_Name") == sys.argv[2]]
print(f"{sys.argv[2]}: {len(vals)} launches, mean {sum(vals) / max(len(vals), 1):.1f}, min {min(vals):.1f}, max {max(vals):.1f}")
PY
  [ -n "$f" ] && grep "attn_sp_fwd_mfma" "$f" | head -12 > gpurun_out/pmc/${c}_attn_rows.csv
done
