#!/bin/bash
# steady-state single-stream kernel summary of the default bench step (top kernels), plus the plain bench line
cd "$(dirname "$0")/.."
O=gpurun_out/profq; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['value'], 'host_enqueue', d.get('host_enqueue_ms'))"
rm -rf /tmp/prof_q
(cd /tmp && MAED_WGRAD_SIDE_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q -o bench -- python "$OLDPWD/bench.py" --steps 6 --warmup 3 --no-cpu-baseline > "$OLDPWD/$O/prof.log" 2>&1)
tr1=$(find /tmp/prof_q -name "*kernel_trace.csv" | head -1)
python scripts/summarize_trace.py "$tr1" $O/kernels_single_stream.csv 4 > /dev/null; head -60 $O/kernels_single_stream.csv | cut -c1-140
