#!/bin/bash
# round 5, call 2: STE twins (bf16x3 forward into fp32 work buffers, bf16 twins saved, bf16 backward) vs the one-plane backward, same box
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5c2; rm -rf $O; mkdir -p $O
for b in bf16x1 bf16 bf16x1 bf16; do
timeout 300 python bench.py --steps 10 --warmup 3 --dtype f32 --f32-matmul bf16x3 --f32-backward $b --no-cpu-baseline 2>$O/err_$b.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$b', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
done
tail -3 $O/err_bf16.log
