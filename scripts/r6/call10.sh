#!/bin/bash
# round 6, call 10: the whole GPU suite at HEAD + cfg5 (long-clip stress) A/B: persistent K-stream kernels on (default heuristics) / off, two interleaved repeats
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r6c10; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest exit: $?" >> $O/pytest_gpu.log
grep -E "passed|failed|Error|error" $O/pytest_gpu.log | tail -n 8
cp gpurun_out/parity_report.txt $O/parity_report_gpu.txt 2>/dev/null
for r in 1 2; do
  for v in 0 1; do
    MAED_SK=$v MAED_TN_SK=$v timeout 600 python bench.py --workload cfg5 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_cfg5_sk${v}_$r.json 2> $O/bench_cfg5_sk${v}_$r.err
    python - <<PY
import json
j = json.loads(open("$O/bench_cfg5_sk${v}_$r.json").read().strip().splitlines()[-1])
print("cfg5 MAED_SK=MAED_TN_SK=$v run $r:", j["ms_per_step"], "ms", j["value"], "clips/s", "roofline", (j.get("roofline") or {}).get("frac"), "nt", (j.get("roofline_nt") or {}).get("frac"))
PY
  done
done
