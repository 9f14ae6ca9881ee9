#!/bin/bash
# 1x1 convolutions on libmaed_hip GEMMs: parity + whole-model tests + phase timings + bench
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --timeout=600 -p no:cacheprovider -k "conv1x1 or backbone or weight_std or train or maed or vit or rccl" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|Error|error|assert" gpurun_out/pytest_gpu.log | tail -n 12
timeout 600 python scripts/diag_step.py > gpurun_out/diag.log 2>&1; echo "diag exit: $?" >> gpurun_out/diag.log; grep -E "fwd\+bwd|train step|exit|Error" gpurun_out/diag.log | grep -vE "#0" | tail -12
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-330
