#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4c13; mkdir -p $O; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$tag.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_$tag.json'));print('$tag', d['ms_per_step'], d.get('host_enqueue_ms'))"; }
timeout 900 python -m pytest tests/test_gpu_tail.py tests/test_gpu_model.py -m gpu -x -q -k "tail or cfg3 or cfg1 or train or golden or bf16" -p no:cacheprovider > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run head_exact MAED_HEAD_X3=0
run head_x3 A=1
run head_exact2 MAED_HEAD_X3=0
run head_x3_2 A=1
