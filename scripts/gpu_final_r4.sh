#!/bin/bash
# round-4 evidence run at HEAD: full GPU test suite, smoke, the default bench line (bf16 headline + roofline + cpu_baseline + parity_mode incl. the mixed mode),
# forward-only, the fp32 modes (bf16x3, backbone bf16x6, bf16x3 forward / bf16 backward), cfg5, the contract's torch.distributed.run line with one rank and forced
# collectives (library communicator = default, torch.distributed beside it), steady-state rocprofv3 summaries (side stream on / single stream), PMC traffic passes
# stamped with the kernel-source hash, the vendor GEMM yardstick, the GroupNorm-backward micro-benchmark.
# Everything lands under gpurun_out/final/ -- scripts/collect_evidence_r4.sh copies what should be judged to profiles/r04_*.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/final; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest exit: $?" >> $O/pytest_gpu.log
grep -E "passed|failed" $O/pytest_gpu.log | tail -n 2
cp gpurun_out/parity_report.txt $O/parity_report_gpu.txt 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit: $?" >> $O/smoke.log; tail -n 2 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err; echo "bench exit: $?" >> $O/bench_train.err; cut -c1-260 $O/bench_train.json
timeout 300 python bench.py --steps 20 --warmup 3 --forward-only --no-cpu-baseline > $O/bench_forward.json 2>/dev/null; cut -c1-200 $O/bench_forward.json
timeout 600 python bench.py --steps 10 --warmup 2 --dtype f32 --f32-matmul bf16x3 --no-cpu-baseline > $O/bench_train_f32_bf16x3.json 2>/dev/null; cut -c1-200 $O/bench_train_f32_bf16x3.json
timeout 600 python bench.py --steps 10 --warmup 2 --dtype f32 --f32-matmul bf16x3 --f32-backward bf16x1 --no-cpu-baseline > $O/bench_train_f32_bf16x3_fwd_bf16x1_bwd.json 2>/dev/null; cut -c1-200 $O/bench_train_f32_bf16x3_fwd_bf16x1_bwd.json
timeout 600 python bench.py --steps 10 --warmup 2 --dtype f32 --f32-matmul bf16x3 --backbone-f32-matmul bf16x6 --no-cpu-baseline > $O/bench_train_f32_bf16x3_backbone_x6.json 2>/dev/null; cut -c1-200 $O/bench_train_f32_bf16x3_backbone_x6.json
for comm in direct torch; do
MAED_COMM=$comm MAED_FORCE_COLLECTIVES=1 MAED_WS_PER_STAGE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_torchrun_world1_forced_collectives_$comm.json 2> $O/bench_torchrun_$comm.err; echo "torchrun bench ($comm) exit: $?"; cut -c1-200 $O/bench_torchrun_world1_forced_collectives_$comm.json
done
timeout 600 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg5.json 2>/dev/null; cut -c1-200 $O/bench_cfg5.json
timeout 300 python scripts/gemm_vs_vendor.py 30 > $O/gemm_vs_vendor.txt 2>&1; tail -3 $O/gemm_vs_vendor.txt | cut -c1-200
timeout 300 python scripts/gn_bwd_micro.py 20 > $O/gn_bwd_onepass_micro.txt 2>&1; tail -2 $O/gn_bwd_onepass_micro.txt | cut -c1-250
timeout 300 python scripts/gemm_micro.py 30 all 0 > $O/gemm_micro.txt 2>&1; grep "gemm " $O/gemm_micro.txt | cut -c1-110
timeout 200 python scripts/attn_long_micro.py > $O/attn_long_micro.txt 2>&1; tail -8 $O/attn_long_micro.txt
# steady-state kernel summaries: bf16 (side stream on), bf16 single stream
rm -rf /tmp/prof_out
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out -o bench -- python "$OLDPWD/bench.py" --steps 6 --warmup 3 --no-cpu-baseline > "$OLDPWD/$O/prof.log" 2>&1)
tr=$(find /tmp/prof_out -name "*kernel_trace.csv" | head -1)
python scripts/summarize_trace.py "$tr" $O/rocprofv3_steady_state_kernels.csv 4 && head -30 $O/rocprofv3_steady_state_kernels.csv | cut -c1-150
cp $(find /tmp/prof_out -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_incl_warmup.csv 2>/dev/null
bash scripts/gpu_pmc.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc/traffic.json $O/traffic.json 2>/dev/null; tail -12 $O/pmc.log
rm -rf /tmp/prof_out1
(cd /tmp && MAED_WGRAD_SIDE_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out1 -o bench -- python "$OLDPWD/bench.py" --steps 6 --warmup 3 --no-cpu-baseline > "$OLDPWD/$O/prof_single_stream.log" 2>&1)
tr1=$(find /tmp/prof_out1 -name "*kernel_trace.csv" | head -1)
python scripts/summarize_trace.py "$tr1" $O/rocprofv3_steady_state_kernels_single_stream.csv 4 > /dev/null && python scripts/group_rooflines.py $O/rocprofv3_steady_state_kernels_single_stream.csv $O/traffic.json | head -60
# the default bench line once more WITH the PMC traffic of this very build in place (bench.py quotes traffic only for a matching source hash)
mkdir -p profiles/r04_pmc && cp $O/traffic.json profiles/r04_pmc/traffic.json
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_train_with_traffic.json 2> $O/bench_train_with_traffic.err; cut -c1-200 $O/bench_train_with_traffic.json
cp profiles/r04_pmc/traffic.json $O/traffic_stamped.json
