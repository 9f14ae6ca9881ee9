#!/bin/bash
# LDS-shuffled GEMM epilogue + TN split default: GEMM/conv parity, micro timings, bench
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --timeout=600 -p no:cacheprovider -k "gemm or linear or conv1x1 or block or backbone or mlp or cfg1" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|Error|error|assert" gpurun_out/pytest_gpu.log | tail -n 12
timeout 300 python scripts/gemm_micro.py 30 all 0 2>&1 | grep gemm | cut -c1-110 | tee gpurun_out/gemm_micro.log
timeout 300 python scripts/conv1x1_micro.py 10 2>&1 | grep -E "totals" | tee -a gpurun_out/gemm_micro.log
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-250
