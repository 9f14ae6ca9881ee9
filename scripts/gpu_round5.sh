#!/bin/bash
# decoder tail backward + fused loss on the real library: new parity tests, whole-model gradient test, smoke, phase timings
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt
timeout 900 python -m pytest tests/test_gpu_tail.py tests/test_gpu_model.py -m gpu -q --timeout=600 -p no:cacheprovider -k "tail or loss or train or rccl or cfg1" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|Error|error|assert" gpurun_out/pytest_gpu.log | tail -n 15
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?" >> gpurun_out/smoke.log; tail -n 3 gpurun_out/smoke.log
timeout 600 python scripts/diag_step.py > gpurun_out/diag.log 2>&1; echo "diag exit: $?" >> gpurun_out/diag.log; grep -E "diag|exit|Error" gpurun_out/diag.log | grep -vE "#0" | tail -34
