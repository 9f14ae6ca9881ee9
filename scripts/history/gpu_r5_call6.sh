#!/bin/bash
# round 5, call 6: the LDS-DMA + transposing-read weight-gradient kernel against the register-transposing one -- micro A/B, kernel tests, train-step A/B
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5c6; rm -rf $O; mkdir -p $O
timeout 300 python scripts/tn_micro.py 20 3 2>&1 | tee $O/tn_micro.txt | tail -16
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_x3.py -q -x -k "tn or wgrad" -p no:cacheprovider 2>&1 | tail -4 | tee $O/pytest.log
for d in 0 1 0 1; do
MAED_TN_DMA=$d timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MAED_TN_DMA=$d', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
done
