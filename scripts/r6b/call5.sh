#!/bin/bash
# round 6, second session, call 5: does the replayed graph keep the step's three-stream concurrency?  kernel trace of the graph leg
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r6b; mkdir -p $O; export TMPDIR=/tmp
rm -rf /tmp/gl; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/gl -o gl -- python "$OLDPWD/bench.py" --graph-leg --steps 12 --warmup 3 > "$OLDPWD/$O/graph_trace_run.log" 2>&1)
tail -n 2 $O/graph_trace_run.log | cut -c1-300
python scripts/r6b/graph_overlap.py "$(find /tmp/gl -name '*kernel_trace.csv' | head -1)" 20 | tee $O/graph_overlap.txt
