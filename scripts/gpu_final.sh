#!/bin/bash
# evidence run at HEAD: full GPU test suite, smoke, bench (with CPU baseline + parity probe), forward-only and f32-mode bench lines,
# steady-state rocprofv3 summary, PMC traffic passes (stamped with the kernel-source hash), GEMM micro-benchmark, cfg5.
# Everything lands under gpurun_out/final/ -- copy what should be judged to profiles/rNN_*.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/final; mkdir -p $O; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest exit: $?" >> $O/pytest_gpu.log
grep -E "passed|failed" $O/pytest_gpu.log | tail -n 2
cp gpurun_out/parity_report.txt $O/parity_report_gpu.txt 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit: $?" >> $O/smoke.log; tail -n 2 $O/smoke.log
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err; echo "bench exit: $?" >> $O/bench_train.err; cut -c1-260 $O/bench_train.json
timeout 300 python bench.py --steps 20 --warmup 3 --forward-only --no-cpu-baseline > $O/bench_forward.json 2>/dev/null; cut -c1-200 $O/bench_forward.json
timeout 600 python bench.py --steps 5 --warmup 2 --dtype f32 --no-cpu-baseline > $O/bench_train_f32.json 2>/dev/null; cut -c1-200 $O/bench_train_f32.json
# the multi-GPU launch line of the contract with ONE rank and every bucket's all-reduce forced: process group, broadcast, bucketed RCCL
# all-reduce overlapped with backward, barrier, per-stage weight standardisation (the world > 1 default) -- the code path the driver's N = 2/4/8 runs take, on this
# single-GPU box
MAED_FORCE_COLLECTIVES=1 MAED_WS_PER_STAGE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_torchrun_world1_forced_collectives.json 2> $O/bench_torchrun.err; echo "torchrun bench exit: $?"; cut -c1-200 $O/bench_torchrun_world1_forced_collectives.json
timeout 300 python scripts/gemm_micro.py 30 all 0 > $O/gemm_micro.txt 2>&1; grep "gemm " $O/gemm_micro.txt | cut -c1-110
bash scripts/gpu_gemm_variants.sh > /dev/null 2>&1
rm -rf /tmp/prof_out
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out -o bench -- python "$OLDPWD/bench.py" --steps 6 --warmup 3 --no-cpu-baseline > "$OLDPWD/$O/prof.log" 2>&1)
tr=$(find /tmp/prof_out -name "*kernel_trace.csv" | head -1)
python scripts/summarize_trace.py "$tr" $O/rocprofv3_steady_state_kernels.csv 4 && head -40 $O/rocprofv3_steady_state_kernels.csv | cut -c1-150
cp $(find /tmp/prof_out -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_incl_warmup.csv 2>/dev/null
bash scripts/gpu_pmc.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc/traffic.json $O/traffic.json 2>/dev/null; tail -30 $O/pmc.log
# single-stream kernel summary (no concurrency: per-kernel durations add up to the step) -> per-group time, GroupNorm bytes/s; stored next to the PMC traffic
rm -rf /tmp/prof_out1
(cd /tmp && MAED_WGRAD_SIDE_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out1 -o bench -- python "$OLDPWD/bench.py" --steps 6 --warmup 3 --no-cpu-baseline > "$OLDPWD/$O/prof_single_stream.log" 2>&1)
tr1=$(find /tmp/prof_out1 -name "*kernel_trace.csv" | head -1)
python scripts/summarize_trace.py "$tr1" $O/rocprofv3_steady_state_kernels_single_stream.csv 4 > /dev/null && python scripts/group_rooflines.py $O/rocprofv3_steady_state_kernels_single_stream.csv $O/traffic.json | head -40
timeout 600 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg5.json 2>/dev/null; cut -c1-200 $O/bench_cfg5.json
