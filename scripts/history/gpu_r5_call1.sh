#!/bin/bash
# round 5, call 1: where the mixed mode's 38 ms go (steady-state kernel summary, single stream) -- decides what the fp16 engine has to cover
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5c1; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
rm -rf /tmp/prof_out
(cd /tmp && MAED_WGRAD_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out -o bench -- python "$OLDPWD/bench.py" --steps 6 --warmup 3 --dtype f32 --f32-matmul bf16x3 --f32-backward bf16x1 --no-cpu-baseline > "$OLDPWD/$O/prof.log" 2>&1)
tr=$(find /tmp/prof_out -name "*kernel_trace.csv" | head -1)
python scripts/summarize_trace.py "$tr" $O/mixed_steady_state_single_stream.csv 4 && head -70 $O/mixed_steady_state_single_stream.csv | cut -c1-160
tail -3 $O/prof.log | cut -c1-300
