#!/bin/bash
# round 6, call 1: (a) which vendor kernels win the STE shapes (names under rocprofv3 --kernel-trace); (b) today's box baseline of the bench at HEAD
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); O=gpurun_out/r6c1; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
rm -rf /tmp/vk
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/vk -o vk -- python "$ROOT/scripts/r6/vendor_kernels.py" > "$ROOT/$O/vendor_run.log" 2>&1)
tr=$(find /tmp/vk -name "*kernel_trace.csv" | head -1)
head -1 "$tr" > $O/trace_header.txt
python scripts/r6/vendor_kernels_parse.py "$tr" > $O/vendor_kernels.txt 2>$O/parse.err; cat $O/vendor_kernels.txt | cut -c1-400
grep -c . "$tr"; grep -i "fill" "$tr" | head -3 | cut -c1-600
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.err | grep -E "per-step|timed" 
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r6c1/bench.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j.get("host_enqueue_ms"))
PY
