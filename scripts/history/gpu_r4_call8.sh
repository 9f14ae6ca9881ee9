#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4c8; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py tests/test_gpu_paths.py -m gpu -x -q -k "block or attention or cfg3 or cfg5 or train or mix or golden or g1 or g2 or modes" -p no:cacheprovider > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for i in 1 2; do
MAED_ST_FUSED=0 timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_unfused$i.json 2>/dev/null; cut -c1-160 $O/bench_unfused$i.json
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_fused$i.json 2>/dev/null; cut -c1-160 $O/bench_fused$i.json
done
timeout 300 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg5.json 2>/dev/null; cut -c1-200 $O/bench_cfg5.json
MAED_ST_FUSED=0 timeout 300 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg5_unfused.json 2>/dev/null; cut -c1-200 $O/bench_cfg5_unfused.json
