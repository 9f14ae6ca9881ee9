#!/bin/bash
# twin mode, whole train step: fc1's activation as planes + fc2 on the plane kernel (MAED_X3_PLANES = 6 / 5 / 2) against fp32 activation + twin (0); interleaved, twice
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5c22; rm -rf $O; mkdir -p $O
for rep in 1 2; do
for v in 0 6 5 2; do
MAED_X3_PLANES=$v timeout 300 python bench.py --steps 10 --warmup 3 --dtype f32 --f32-matmul bf16x3 --f32-backward bf16 --no-cpu-baseline --no-ddp-rehearsal 2>$O/err_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MAED_X3_PLANES=$v', d['ms_per_step'], 'ms', d['value'], 'clips/s')" | tee -a $O/ab.txt
done
done
timeout 900 python -m pytest tests/test_gpu_parity_mode.py -q -x -k "twin" -p no:cacheprovider 2>&1 | tail -5 | tee $O/pytest.log
