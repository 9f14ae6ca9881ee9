#!/bin/bash
# A/B of one environment knob on the benchmark, interleaved A B A B: bash scripts/gpu_ab.sh MAED_CONV1X1_S2 [pytest -k expression] [A B]   (default values 1 0)
set -u
cd "$(dirname "$0")/.."
knob=$1; sel=${2:-}; A=${3:-1}; B=${4:-0}
if [ -n "$sel" ]; then timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu -x -k "$sel" 2>&1 | grep -E "passed|failed|Error|error" | tail -5; fi
for f in $A $B $A $B; do env $knob=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$knob=$f', d['ms_per_step'], d['value'])"; done
