#!/bin/bash
# round-3 evidence run at HEAD: full GPU test suite, smoke, the default bench line (bf16 headline + roofline + cpu_baseline + parity_mode), forward-only, the fp32 modes
# (exact / bf16x3 / backbone bf16x6), cfg5, the contract's torch.distributed.run line with one rank and forced collectives, steady-state rocprofv3 summaries (side stream on /
# single stream; bf16 and the fp32-accurate mode), PMC traffic passes stamped with the kernel-source hash, micro-benchmarks of the GEMM families.
# Everything lands under gpurun_out/final/ -- scripts/collect_evidence_r3.sh copies what should be judged to profiles/r03_*.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/final; mkdir -p $O; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest exit: $?" >> $O/pytest_gpu.log
grep -E "passed|failed" $O/pytest_gpu.log | tail -n 2
cp gpurun_out/parity_report.txt $O/parity_report_gpu.txt 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit: $?" >> $O/smoke.log; tail -n 2 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err; echo "bench exit: $?" >> $O/bench_train.err; cut -c1-260 $O/bench_train.json
timeout 300 python bench.py --steps 20 --warmup 3 --forward-only --no-cpu-baseline > $O/bench_forward.json 2>/dev/null; cut -c1-200 $O/bench_forward.json
timeout 600 python bench.py --steps 5 --warmup 2 --dtype f32 --f32-matmul exact --no-cpu-baseline > $O/bench_train_f32_exact.json 2>/dev/null; cut -c1-200 $O/bench_train_f32_exact.json
timeout 600 python bench.py --steps 10 --warmup 2 --dtype f32 --f32-matmul bf16x3 --no-cpu-baseline > $O/bench_train_f32_bf16x3.json 2>/dev/null; cut -c1-200 $O/bench_train_f32_bf16x3.json
timeout 600 python bench.py --steps 10 --warmup 2 --dtype f32 --f32-matmul bf16x3 --backbone-f32-matmul bf16x6 --no-cpu-baseline > $O/bench_train_f32_bf16x3_backbone_x6.json 2>/dev/null; cut -c1-200 $O/bench_train_f32_bf16x3_backbone_x6.json
MAED_FORCE_COLLECTIVES=1 MAED_WS_PER_STAGE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_torchrun_world1_forced_collectives.json 2> $O/bench_torchrun.err; echo "torchrun bench exit: $?"; cut -c1-200 $O/bench_torchrun_world1_forced_collectives.json
timeout 600 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg5.json 2>/dev/null; cut -c1-200 $O/bench_cfg5.json
timeout 300 python scripts/gemm_micro.py 30 all 0 > $O/gemm_micro.txt 2>&1; grep "gemm " $O/gemm_micro.txt | cut -c1-110
timeout 600 python scripts/x3_micro.py 20 > $O/x3_micro.txt 2>&1; cut -c1-200 $O/x3_micro.txt
# steady-state kernel summaries: bf16 (side stream on), bf16 single stream, fp32-accurate mode single stream
rm -rf /tmp/prof_out
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out -o bench -- python "$OLDPWD/bench.py" --steps 6 --warmup 3 --no-cpu-baseline > "$OLDPWD/$O/prof.log" 2>&1)
tr=$(find /tmp/prof_out -name "*kernel_trace.csv" | head -1)
python scripts/summarize_trace.py "$tr" $O/rocprofv3_steady_state_kernels.csv 4 && head -30 $O/rocprofv3_steady_state_kernels.csv | cut -c1-150
cp $(find /tmp/prof_out -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_incl_warmup.csv 2>/dev/null
bash scripts/gpu_pmc.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc/traffic.json $O/traffic.json 2>/dev/null; tail -12 $O/pmc.log
rm -rf /tmp/prof_out1
(cd /tmp && MAED_WGRAD_SIDE_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out1 -o bench -- python "$OLDPWD/bench.py" --steps 6 --warmup 3 --no-cpu-baseline > "$OLDPWD/$O/prof_single_stream.log" 2>&1)
tr1=$(find /tmp/prof_out1 -name "*kernel_trace.csv" | head -1)
python scripts/summarize_trace.py "$tr1" $O/rocprofv3_steady_state_kernels_single_stream.csv 4 > /dev/null && python scripts/group_rooflines.py $O/rocprofv3_steady_state_kernels_single_stream.csv $O/traffic.json | head -40
rm -rf /tmp/prof_out2
(cd /tmp && MAED_WGRAD_SIDE_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out2 -o bench -- python "$OLDPWD/bench.py" --steps 4 --warmup 2 --dtype f32 --f32-matmul bf16x3 --backbone-f32-matmul bf16x6 --no-cpu-baseline > "$OLDPWD/$O/prof_f32.log" 2>&1)
tr2=$(find /tmp/prof_out2 -name "*kernel_trace.csv" | head -1)
python scripts/summarize_trace.py "$tr2" $O/rocprofv3_steady_state_kernels_f32_accurate_single_stream.csv 3 | head -3
# the default bench line once more WITH the PMC traffic of this very build in place (bench.py quotes traffic only for a matching source hash)
mkdir -p profiles/r03_pmc && cp $O/traffic.json profiles/r03_pmc/traffic.json
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_train_with_traffic.json 2> $O/bench_train_with_traffic.err; cut -c1-200 $O/bench_train_with_traffic.json
cp profiles/r03_pmc/traffic.json $O/traffic_stamped.json
