#!/bin/bash
# the GPU test suite + smoke once more at HEAD (tests changed, kernels did not): replaces the evidence run's pytest log / parity report
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/final; mkdir -p $O
rm -f gpurun_out/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest exit: $?" >> $O/pytest_gpu.log
grep -E "passed|failed" $O/pytest_gpu.log | tail -n 2
cp gpurun_out/parity_report.txt $O/parity_report_gpu.txt 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit: $?" >> $O/smoke.log; tail -n 2 $O/smoke.log
