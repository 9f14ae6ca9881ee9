#!/bin/bash
# same-box A/B of the default bench line under two settings of one environment variable: usage gpu_ab_env.sh VAR A B [rounds]
cd "$(dirname "$0")/.."
VAR=$1; A=$2; B=$3; R=${4:-2}
for r in $(seq $R); do for v in $A $B; do
env $VAR=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['ms_per_step'], d['value'])"
done; done
