#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4c5; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python bench.py --steps 10 --warmup 2 --dtype f32 --f32-matmul bf16x3 --no-cpu-baseline > $O/bench_x3_train.json 2>/dev/null; cut -c1-200 $O/bench_x3_train.json
timeout 600 python bench.py --steps 10 --warmup 2 --dtype f32 --f32-matmul bf16x3 --forward-only --no-cpu-baseline > $O/bench_x3_fwd.json 2>/dev/null; cut -c1-200 $O/bench_x3_fwd.json
timeout 600 python bench.py --steps 10 --warmup 2 --forward-only --no-cpu-baseline > $O/bench_bf16_fwd.json 2>/dev/null; cut -c1-200 $O/bench_bf16_fwd.json
