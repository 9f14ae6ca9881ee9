#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4c4; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_parity_mode.py -m gpu -x -q -k "groupnorm or backbone or resnet or cfg3 or train or parity" -p no:cacheprovider > $O/pytest_backbone.log 2>&1; tail -3 $O/pytest_backbone.log
for i in 1 2; do
MAED_GN_BWD_ONEPASS=0 timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_twopass$i.json 2>/dev/null; cut -c1-200 $O/bench_twopass$i.json
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_onepass$i.json 2>/dev/null; cut -c1-200 $O/bench_onepass$i.json
done
