#!/bin/bash
# twin mode, whole train step: plane storage off / fc2 only / fc2 + LayerNorm planes with qkv and fc1 on the plane kernel; interleaved, twice; then the twin GPU tests
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5c23; rm -rf $O; mkdir -p $O
for rep in 1 2; do
for v in "0 0" "6 0" "6 1"; do
set -- $v
MAED_X3_PLANES=$1 MAED_X3_PLANES_LN=$2 timeout 300 python bench.py --steps 10 --warmup 3 --dtype f32 --f32-matmul bf16x3 --f32-backward bf16 --no-cpu-baseline --no-ddp-rehearsal 2>$O/err_$1_$2.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MAED_X3_PLANES=$1 MAED_X3_PLANES_LN=$2', d['ms_per_step'], 'ms', d['value'], 'clips/s')" | tee -a $O/ab.txt
done
done
timeout 900 python -m pytest tests/test_gpu_parity_mode.py -q -x -k "twin" -p no:cacheprovider 2>&1 | tail -5 | tee $O/pytest.log
