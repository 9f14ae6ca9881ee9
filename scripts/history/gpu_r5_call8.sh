#!/bin/bash
# round 5, call 8: weight-gradient kernel choice in situ (0 = register-transposing everywhere, 1 = LDS-DMA kernel where N, K >= 512, 2 = LDS-DMA kernel everywhere)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5c8; rm -rf $O; mkdir -p $O
for d in 0 1 2 0 1 2; do
MAED_TN_DMA=$d timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MAED_TN_DMA=$d', d['ms_per_step'], d['value'], 'wgrad', d['roofline_wgrad']['avg_us'], d['roofline_wgrad']['frac'], 'roofline', d['roofline']['frac'])" | tee -a $O/ab.txt
done
