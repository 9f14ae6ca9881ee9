#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4c12; mkdir -p $O; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$tag.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_$tag.json'));print('$tag', d['ms_per_step'], d.get('host_enqueue_ms'))"; }
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "embed or weight_std or backbone or resnet or cfg3 or cfg1 or g4 or golden" -p no:cacheprovider > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run ws_wave MAED_WS_TILED=0
run ws_tiled A=1
run ws_wave2 MAED_WS_TILED=0
run ws_tiled2 A=1
