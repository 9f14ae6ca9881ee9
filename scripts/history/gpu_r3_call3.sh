#!/bin/bash
# round 3, third GPU call: software-pipelined split kernels -- tests, micro-benchmark, step times
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3c3; mkdir -p $O; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt
timeout 900 python -m pytest tests/test_gpu_x3.py -m gpu -q --timeout=600 -p no:cacheprovider > $O/pytest_x3.log 2>&1; echo "pytest exit: $?" >> $O/pytest_x3.log
tail -n 4 $O/pytest_x3.log
timeout 600 python scripts/x3_micro.py 20 ${MICRO_GROUPS:-nt,tn,conv} > $O/x3_micro.txt 2>&1; cat $O/x3_micro.txt | cut -c1-200
for m in "bf16x3" "bf16x3 --backbone-f32-matmul bf16x6"; do
  tag=$(echo $m | tr -d ' -'); timeout 600 python bench.py --steps 5 --warmup 2 --dtype f32 --f32-matmul $m --no-cpu-baseline > $O/bench_f32_$tag.json 2> $O/bench_f32_$tag.err; echo "bench $m exit: $?"; cut -c1-220 $O/bench_f32_$tag.json
done
