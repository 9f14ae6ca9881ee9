#!/bin/bash
# round 3, second GPU call: attention on the split kernels, STE block TN path, mixed-precision backbone; micro-benchmarks of every split kernel
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3c2; mkdir -p $O; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt
timeout 900 python -m pytest tests/test_gpu_x3.py -m gpu -q --timeout=600 -p no:cacheprovider > $O/pytest_x3.log 2>&1; echo "pytest exit: $?" >> $O/pytest_x3.log
tail -n 5 $O/pytest_x3.log
cp gpurun_out/parity_report.txt $O/parity_report_x3.txt 2>/dev/null
timeout 600 python scripts/x3_micro.py 20 > $O/x3_micro.txt 2>&1; cat $O/x3_micro.txt | cut -c1-200
timeout 900 python scripts/x3_probe.py bf16x3 "bf16x3+bb:bf16x6" bf16x6 > $O/x3_probe.txt 2>&1; echo "probe exit: $?"; grep "mode\|median" $O/x3_probe.txt | tail -40
for m in "bf16x3" "bf16x3 --backbone-f32-matmul bf16x6" "bf16x6"; do
  tag=$(echo $m | tr -d ' -'); timeout 600 python bench.py --steps 5 --warmup 2 --dtype f32 --f32-matmul $m --no-cpu-baseline > $O/bench_f32_$tag.json 2> $O/bench_f32_$tag.err; echo "bench $m exit: $?"; cut -c1-220 $O/bench_f32_$tag.json
done
rm -rf /tmp/prof_x3
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x3 -o bench -- python "$OLDPWD/bench.py" --steps 4 --warmup 2 --dtype f32 --f32-matmul bf16x3 --backbone-f32-matmul bf16x6 --no-cpu-baseline > "$OLDPWD/$O/prof_x3.log" 2>&1)
tr=$(find /tmp/prof_x3 -name "*kernel_trace.csv" | head -1)
python scripts/summarize_trace.py "$tr" $O/rocprofv3_steady_state_kernels_f32_mixed.csv 3 && head -45 $O/rocprofv3_steady_state_kernels_f32_mixed.csv | cut -c1-160
