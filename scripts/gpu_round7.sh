#!/bin/bash
# GroupNorm backward-reduce unroll: parity tests + steady-state profile with the ordered kernel sequence of one step
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=600 -p no:cacheprovider -k "groupnorm or weight_std or backbone" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|Error|error|assert" gpurun_out/pytest_gpu.log | tail -n 8
bash scripts/gpu_prof.sh > gpurun_out/prof_stdout.log 2>&1; head -3 gpurun_out/prof_stdout.log | cut -c1-200; grep -E "gn_bwd_reduce|gn_bwd_apply" gpurun_out/prof/steady_state_kernels.csv | cut -c1-120
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-330
