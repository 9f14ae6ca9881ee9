#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/pytest_gpu.log | tail -25
timeout 600 python scripts/diag_step.py > gpurun_out/diag.log 2>&1; echo "diag exit: $?" >> gpurun_out/diag.log; grep -E "diag|exit|Error" gpurun_out/diag.log | grep -vE "#1" | tail -30
