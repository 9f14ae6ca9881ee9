#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3c13; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python scripts/x3_micro.py 20 tn all bf16,bf16x3 2>&1 | grep "^tn" | cut -c1-150 | tee $O/tn_new_splits.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; python - <<PY
import json
d=json.loads(open("$O/bench_bf16.json").read().strip().splitlines()[-1])
print("bf16", d["ms_per_step"], d["step_time"]["median_ms"], "host", d.get("host_enqueue_ms"), json.dumps(d["roofline"])[:300])
PY
rm -rf /tmp/prof_x3
(cd /tmp && MAED_WGRAD_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x3 -o bench -- python "$OLDPWD/bench.py" --steps 4 --warmup 2 --dtype f32 --f32-matmul bf16x3 --backbone-f32-matmul bf16x6 --no-cpu-baseline > "$OLDPWD/$O/prof_x3.log" 2>&1)
tr=$(find /tmp/prof_x3 -name "*kernel_trace.csv" | head -1)
python scripts/summarize_trace.py "$tr" $O/rocprofv3_steady_state_kernels_f32_mixed_single_stream.csv 3 && head -42 $O/rocprofv3_steady_state_kernels_f32_mixed_single_stream.csv | cut -c1-150
