#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4c7; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python bench.py --steps 10 --warmup 2 --dtype f32 --f32-matmul bf16x3 --no-cpu-baseline > $O/bench_x3_train.json 2>/dev/null; cut -c1-200 $O/bench_x3_train.json
timeout 600 python bench.py --steps 10 --warmup 2 --dtype f32 --f32-matmul bf16x3 --f32-backward bf16x1 --no-cpu-baseline > $O/bench_x3_fwd_x1_bwd.json 2> $O/bench_x3_fwd_x1_bwd.err; cut -c1-200 $O/bench_x3_fwd_x1_bwd.json; tail -3 $O/bench_x3_fwd_x1_bwd.err
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_x3.py tests/test_gpu_parity_mode.py -m gpu -x -q -k "cfg5 or x3 or parity" -p no:cacheprovider > $O/pytest.log 2>&1; tail -5 $O/pytest.log
