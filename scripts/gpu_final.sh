#!/bin/bash
# evidence run: full GPU test suite, smoke, bench (with CPU baseline), steady-state rocprofv3 summary
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
tail -n 3 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?" >> gpurun_out/smoke.log; tail -n 2 gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit: $?" >> gpurun_out/bench.err; tail -n 4 gpurun_out/bench.err; cat gpurun_out/bench.log
timeout 300 python bench.py --steps 20 --warmup 3 --forward-only --no-cpu-baseline > gpurun_out/bench_fwd.log 2>/dev/null; cat gpurun_out/bench_fwd.log | cut -c1-400
timeout 300 python scripts/gemm_micro.py 30 all 0 > gpurun_out/gemm_micro.log 2>&1; grep -E "impl 0" gpurun_out/gemm_micro.log | cut -c1-100
bash scripts/gpu_prof.sh > gpurun_out/prof_stdout.log 2>&1; head -5 gpurun_out/prof_stdout.log | cut -c1-200
timeout 600 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_cfg5.log 2> gpurun_out/bench_cfg5.err; cut -c1-200 gpurun_out/bench_cfg5.log
