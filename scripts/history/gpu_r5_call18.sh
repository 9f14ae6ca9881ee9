#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5c18; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_mode.py -q -x -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 --dtype f32 --f32-matmul bf16x3 --f32-backward bf16 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('twin', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
