#!/bin/bash
# rocprofv3 kernel trace of the bench -> steady-state per-kernel summary (small CSV) under gpurun_out/prof/
set -u
cd "$(dirname "$0")/.."
ROOT=$(pwd)
mkdir -p gpurun_out/prof; export TMPDIR=/tmp
rm -rf /tmp/prof_out
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out -o bench -- python "$ROOT/bench.py" --steps 6 --warmup 3 --no-cpu-baseline ${BENCH_EXTRA:-} > "$ROOT/gpurun_out/prof.log" 2>&1); echo "prof exit: $?" >> gpurun_out/prof.log
tail -n 2 gpurun_out/prof.log
tr=$(find /tmp/prof_out -name "*kernel_trace.csv" | head -1)
python scripts/summarize_trace.py "$tr" gpurun_out/prof/steady_state_kernels.csv 4 && head -70 gpurun_out/prof/steady_state_kernels.csv
cp $(find /tmp/prof_out -name "*kernel_stats.csv" | head -1) gpurun_out/prof/ 2>/dev/null
