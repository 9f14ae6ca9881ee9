#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4c14; mkdir -p $O; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$tag.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_$tag.json'));print('$tag', d['ms_per_step'], d.get('host_enqueue_ms'))"; }
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -x -q -k "backbone or resnet or cfg3 or cfg1 or cfg2 or golden or g5" -p no:cacheprovider > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run stem_aten MAED_STEM_INPUT=0
run stem_own A=1
run stem_aten2 MAED_STEM_INPUT=0
run stem_own2 A=1
