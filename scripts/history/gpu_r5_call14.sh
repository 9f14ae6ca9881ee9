#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5c14; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for fb in 16 0; do
rm -rf /tmp/prof_out
(cd /tmp && MAED_LBS_FB=$fb timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out -o lbs -- python "$OLDPWD/scripts/lbs_micro.py" 20 > "$OLDPWD/$O/prof.log" 2>&1)
f=$(find /tmp/prof_out -name "*kernel_stats.csv" | head -1)
python - "$f" $fb <<'PY' | tee -a $O/lbs_kernels.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "lbs" in r["Name"]:
        print(f"MAED_LBS_FB={sys.argv[2]}  {r['Name'][:40]:40s} calls {r['Calls']:>4s}  avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
