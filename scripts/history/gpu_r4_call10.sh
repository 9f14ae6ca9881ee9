#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4c10; mkdir -p $O; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$tag.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_$tag.json'));print('$tag', d['ms_per_step'], d.get('host_enqueue_ms'))"; }
run base A=1
run side0 MAED_WGRAD_SIDE_STREAM=0
run gn2pass MAED_GN_BWD_ONEPASS=0
run gn2pass_side0 MAED_GN_BWD_ONEPASS=0 MAED_WGRAD_SIDE_STREAM=0
run base2 A=1
