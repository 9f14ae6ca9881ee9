#!/bin/bash
# rocprofv3 kernel trace of the bench (no CPU baseline) + the bench itself with the CPU baseline
set -u
cd "$(dirname "$0")/.."
ROOT=$(pwd)
mkdir -p gpurun_out; export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/prof" -o bench -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$ROOT/gpurun_out/prof.log" 2>&1); echo "prof exit: $?" >> gpurun_out/prof.log
tail -n 3 gpurun_out/prof.log
find gpurun_out/prof -type f | head -10
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -45 "$f"
find gpurun_out/prof -name "*kernel_trace.csv" -size +30M -delete   # keep the merge under the 64 MiB cap
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit: $?" >> gpurun_out/bench.err; tail -n 6 gpurun_out/bench.err; tail -n 2 gpurun_out/bench.log
