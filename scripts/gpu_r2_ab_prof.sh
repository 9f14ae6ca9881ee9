#!/bin/bash
# round 2, call 2: per-kernel A/B of the opt-in switches under rocprofv3 (steady-state summaries), and the GPU suite with every switch on
set -u
cd "$(dirname "$0")/.."
ROOT=$(pwd); mkdir -p gpurun_out/r02; export TMPDIR=/tmp
OPTIN="MAED_GN_DEFER_AFFINE=1 MAED_LN_DEFER_AFFINE=1 MAED_TAIL_PARALLEL=1 MAED_TM_BWD_L32=1 MAED_TM_BWD_WIDE_REGS=1 MAED_WS_PER_STAGE=1"
prof() {  # name, env...
  name=$1; shift
  rm -rf /tmp/prof_$name
  (cd /tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o bench -- python "$ROOT/bench.py" --steps 6 --warmup 3 --no-cpu-baseline > "$ROOT/gpurun_out/r02/prof_$name.log" 2>&1)
  tr=$(find /tmp/prof_$name -name "*kernel_trace.csv" | head -1)
  python scripts/summarize_trace.py "$tr" gpurun_out/r02/steady_$name.csv 4
}
prof base MAED_CONV3X3=own
prof optin MAED_CONV3X3=own $OPTIN
env MAED_CONV3X3=own $OPTIN MAED_RUN_UNVERIFIED_GPU_TESTS=1 timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r02/pytest_all_optin.txt
