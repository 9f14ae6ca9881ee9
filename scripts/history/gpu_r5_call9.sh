#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5c9; rm -rf $O; mkdir -p $O
for b in bf16 bf16 bf16x1; do
timeout 300 python bench.py --steps 10 --warmup 3 --dtype f32 --f32-matmul bf16x3 --f32-backward $b --no-cpu-baseline 2>$O/err_$b.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$b', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
done
timeout 900 python -m pytest tests/test_gpu_parity_mode.py tests/test_gpu_kernels.py -q -x -k "mixed_mode or twin_mode or frame_barrier or gemm_nt" -p no:cacheprovider 2>&1 | tail -5 | tee $O/pytest.log
