#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4c16; mkdir -p $O; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$tag.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_$tag.json'));print('$tag', d['ms_per_step'], d.get('host_enqueue_ms'))"; }
run legacy MAED_ST_FUSED=2
run piggy_only A=1
run both MAED_ST_FUSED=4
run legacy2 MAED_ST_FUSED=2
run piggy_only2 A=1
run both2 MAED_ST_FUSED=4
