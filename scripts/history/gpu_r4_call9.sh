#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4c9; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_paths.py -m gpu -x -q -k "conv or backbone or resnet or cfg3 or train or groupnorm" -p no:cacheprovider > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for i in 1 2; do
MAED_CONV3X3_S2=0 timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_miopen_s2_$i.json 2>/dev/null; cut -c1-160 $O/bench_miopen_s2_$i.json
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_own_s2_$i.json 2>/dev/null; cut -c1-160 $O/bench_own_s2_$i.json
done
