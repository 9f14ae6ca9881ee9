#!/bin/bash
# One GPU session: parity tests (all, no -x, per-kernel error report), smoke, bench, rocprof summary.
# usage (from the repo root on the GPU box): bash scripts/gpu_round.sh [tests|bench|prof|all]
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
what="${1:-all}"
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt
if [[ "$what" == "tests" || "$what" == "all" ]]; then
  timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
  tail -n 60 gpurun_out/pytest_gpu.log
fi
if [[ "$what" == "bench" || "$what" == "all" ]]; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?" >> gpurun_out/smoke.log; tail -n 5 gpurun_out/smoke.log
  timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench exit: $?" >> gpurun_out/bench.log; tail -n 5 gpurun_out/bench.log
fi
if [[ "$what" == "prof" || "$what" == "all" ]]; then
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof.log" 2>&1)
  echo "prof exit: $?" >> gpurun_out/prof.log
  find gpurun_out/prof -name "*kernel_stats*" | head -3
fi
