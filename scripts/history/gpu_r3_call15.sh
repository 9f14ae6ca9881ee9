#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3c15; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python scripts/x3_micro.py 30 tn,conv all bf16 2>&1 | grep "^tn\|^conv" | cut -c1-160 | tee $O/tn_bf16_buffer_loads.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "tn or wgrad or conv" --timeout=600 -p no:cacheprovider 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; python - <<PY
import json
d=json.loads(open("$O/bench_bf16.json").read().strip().splitlines()[-1])
print("bf16", d["ms_per_step"], d["step_time"]["median_ms"], "host", d.get("host_enqueue_ms"), json.dumps(d["roofline"])[:330])
PY
