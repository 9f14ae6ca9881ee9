#!/bin/bash
# single-stream steady-state rocprofv3 kernel summary of the default bench line -> gpurun_out/r4prof/steady_single_stream.csv
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4prof; mkdir -p $O; export TMPDIR=/tmp
rm -rf /tmp/prof_out1
(cd /tmp && MAED_WGRAD_SIDE_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out1 -o bench -- python "$OLDPWD/bench.py" --steps 6 --warmup 3 --no-cpu-baseline > "$OLDPWD/$O/prof_single_stream.log" 2>&1)
tr1=$(find /tmp/prof_out1 -name "*kernel_trace.csv" | head -1)
python scripts/summarize_trace.py "$tr1" $O/steady_single_stream.csv 4 > /dev/null; head -60 $O/steady_single_stream.csv | cut -c1-130
