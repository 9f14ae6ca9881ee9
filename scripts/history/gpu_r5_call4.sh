#!/bin/bash
# round 5, call 4: twins written by the GroupNorm forward instead of cast passes -- A/B + parity tests + profile
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5c4; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for b in bf16 bf16x1 bf16; do
timeout 300 python bench.py --steps 10 --warmup 3 --dtype f32 --f32-matmul bf16x3 --f32-backward $b --no-cpu-baseline 2>$O/err_$b.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$b', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
done
grep -i "error\|Traceback" -A5 $O/err_bf16.log | head -30
timeout 900 python -m pytest tests/test_gpu_parity_mode.py -q -x -k "mixed_mode or twin_mode" -p no:cacheprovider 2>&1 | tail -5 | tee $O/pytest.log
cp gpurun_out/parity_report.txt $O/parity_report.txt 2>/dev/null
rm -rf /tmp/prof_out
(cd /tmp && MAED_WGRAD_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out -o bench -- python "$OLDPWD/bench.py" --steps 6 --warmup 3 --dtype f32 --f32-matmul bf16x3 --f32-backward bf16 --no-cpu-baseline > "$OLDPWD/$O/prof.log" 2>&1)
tr=$(find /tmp/prof_out -name "*kernel_trace.csv" | head -1)
python scripts/summarize_trace.py "$tr" $O/twin_steady_state_single_stream.csv 4 | tail -1
