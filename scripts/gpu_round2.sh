#!/bin/bash
# second GPU session: re-run the tests that changed, phase diagnostics, bench with progress log
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/pytest_gpu.log | tail -15
timeout 600 python scripts/diag_step.py > gpurun_out/diag.log 2>&1; echo "diag exit: $?" >> gpurun_out/diag.log; grep -E "diag|exit|Error" gpurun_out/diag.log | tail -45
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit: $?" >> gpurun_out/bench.err; tail -n 12 gpurun_out/bench.err; tail -n 2 gpurun_out/bench.log
