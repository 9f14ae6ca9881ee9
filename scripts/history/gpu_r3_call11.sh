#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3c11; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python scripts/x3_micro.py 20 tn,conv all bf16x3,bf16x6 2>&1 | grep "^tn\|^conv" | cut -c1-200 | tee $O/x3_tn_coalesced.txt
timeout 600 python -m pytest tests/test_gpu_x3.py tests/test_gpu_parity_mode.py -m gpu -q --timeout=600 -p no:cacheprovider 2>&1 | tail -2
for m in "bf16x3" "bf16x3 --backbone-f32-matmul bf16x6"; do
  tag=$(echo $m | tr -d ' -'); timeout 600 python bench.py --steps 8 --warmup 2 --dtype f32 --f32-matmul $m --no-cpu-baseline > $O/bench_f32_$tag.json 2> $O/bench_f32_$tag.err; echo "bench $m exit: $?"; cut -c1-200 $O/bench_f32_$tag.json
done
