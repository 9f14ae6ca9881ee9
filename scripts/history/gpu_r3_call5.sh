#!/bin/bash
# round 3: full GPU suite + smoke + default bench (with parity_mode) at the current tree
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3c5; mkdir -p $O; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest exit: $?" >> $O/pytest_gpu.log
grep -E "passed|failed|FAILED|Error" $O/pytest_gpu.log | tail -n 15
cp gpurun_out/parity_report.txt $O/parity_report_gpu.txt 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit: $?" >> $O/smoke.log; tail -n 2 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err; echo "bench exit: $?" >> $O/bench_train.err; cut -c1-260 $O/bench_train.json; tail -3 $O/bench_train.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3c5/bench_train.json").read().strip().splitlines()[-1])
print("parity_mode:", json.dumps(d.get("parity_mode"))[:600])
PY
