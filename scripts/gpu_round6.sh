#!/bin/bash
# cfg5 + direct-RCCL tests, GEMM staging variants (register vs direct global->LDS), cfg5 bench line
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python scripts/gemm_micro.py 30 all 2,3,4 > gpurun_out/gemm_micro.log 2>&1; echo "micro exit: $?" >> gpurun_out/gemm_micro.log; cat gpurun_out/gemm_micro.log | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout=600 -p no:cacheprovider -k "cfg5 or direct_rccl" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|Error|error|assert" gpurun_out/pytest_gpu.log | tail -n 15
timeout 600 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_cfg5.log 2> gpurun_out/bench_cfg5.err; echo "bench cfg5 exit: $?" >> gpurun_out/bench_cfg5.err; tail -n 3 gpurun_out/bench_cfg5.err; cut -c1-700 gpurun_out/bench_cfg5.log
