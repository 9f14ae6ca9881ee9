#!/bin/bash
# round 4, call 2: one-pass GroupNorm backward -- micro-benchmark vs the two-pass kernels, the backbone GPU tests, bench A/B
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4c2; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python scripts/gn_bwd_micro.py 20 > $O/gn_bwd_micro.txt 2>&1; cat $O/gn_bwd_micro.txt | tail -16
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "groupnorm or backbone or resnet or cfg3 or train" -p no:cacheprovider > $O/pytest_backbone.log 2>&1; tail -3 $O/pytest_backbone.log
MAED_GN_BWD_ONEPASS=0 timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_twopass.json 2>/dev/null; cut -c1-200 $O/bench_twopass.json
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_onepass.json 2>/dev/null; cut -c1-200 $O/bench_onepass.json
MAED_GN_BWD_ONEPASS=0 timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_twopass2.json 2>/dev/null; cut -c1-200 $O/bench_twopass2.json
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_onepass2.json 2>/dev/null; cut -c1-200 $O/bench_onepass2.json
