#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3c7; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python scripts/x3_probe.py "bf16x3+bb:bf16x6" > $O/x3_probe_wgrad_x3.txt 2>&1; grep "mode\|median\|out theta" $O/x3_probe_wgrad_x3.txt | tail -20
timeout 600 python bench.py --steps 8 --warmup 2 --dtype f32 --f32-matmul bf16x3 --backbone-f32-matmul bf16x6 --no-cpu-baseline > $O/bench_f32_mixed.json 2> $O/bench_f32_mixed.err; cut -c1-200 $O/bench_f32_mixed.json
timeout 600 python -m pytest tests/test_gpu_parity_mode.py -m gpu -q --timeout=600 -p no:cacheprovider 2>&1 | tail -3
