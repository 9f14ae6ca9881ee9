#!/bin/bash
# round-6 evidence run at HEAD: full GPU test suite, smoke, the default bench line (bf16 headline + roofline + cpu_baseline + parity_mode incl. the twin mode +
# ddp.rehearsal), forward-only, the fp32 modes (bf16x3, one-plane backward, twins), cfg5 in bf16 and in the twin mode, steady-state rocprofv3 summaries (bf16 side
# stream on / single stream; twin mode single stream), PMC traffic passes stamped with the kernel-source hash, the vendor GEMM yardstick, GEMM / TN / attention micro.
# + round 6: the persistent K-stream GEMMs' micro-benchmarks (scripts/r6/sk_micro.py, tn_sk_micro.py) and the vendor library's kernel names at the STE shapes.
# Everything lands under gpurun_out/final/ -- scripts/collect_evidence_r6.sh copies what should be judged to profiles/r06_*.
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
timeout 600 python bench.py --steps 10 --warmup 2 --dtype f32 --f32-matmul bf16x3 --f32-backward bf16 --no-cpu-baseline > $O/bench_train_f32_bf16x3_fwd_bf16_twin_bwd.json 2>/dev/null; cut -c1-200 $O/bench_train_f32_bf16x3_fwd_bf16_twin_bwd.json
timeout 600 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg5.json 2>/dev/null; cut -c1-200 $O/bench_cfg5.json
timeout 600 python bench.py --workload cfg5 --steps 5 --warmup 2 --dtype f32 --f32-matmul bf16x3 --f32-backward bf16 --no-cpu-baseline > $O/bench_cfg5_twin.json 2>/dev/null; cut -c1-200 $O/bench_cfg5_twin.json
timeout 400 python bench.py --graph-leg --steps 20 --warmup 3 > $O/graph_leg.json 2> $O/graph_leg.err; cut -c1-400 $O/graph_leg.json
timeout 300 python -m pytest tests/test_gpu_graph.py -m gpu -q -s -p no:cacheprovider > $O/test_gpu_graph.log 2>&1; tail -n 3 $O/test_gpu_graph.log
timeout 600 python scripts/stress_r6_kernels.py 60 > $O/stress_r6_kernels.txt 2>&1; tail -n 4 $O/stress_r6_kernels.txt | cut -c1-200
timeout 300 python scripts/gemm_vs_vendor.py 30 > $O/gemm_vs_vendor.txt 2>&1; tail -3 $O/gemm_vs_vendor.txt | cut -c1-200
timeout 300 python scripts/gemm_micro.py 30 all 0 > $O/gemm_micro.txt 2>&1; grep "gemm " $O/gemm_micro.txt | cut -c1-110
timeout 300 python scripts/tn_micro.py 20 3 > $O/tn_micro.txt 2>&1; tail -15 $O/tn_micro.txt | cut -c1-200
timeout 200 python scripts/attn_long_micro.py > $O/attn_long_micro.txt 2>&1; tail -8 $O/attn_long_micro.txt
timeout 600 python scripts/r6/sk_micro.py 20 10 > $O/sk_micro.txt 2>&1; tail -15 $O/sk_micro.txt | cut -c1-200
timeout 600 python scripts/r6/tn_sk_micro.py 20 10 > $O/tn_sk_micro.txt 2>&1; tail -14 $O/tn_sk_micro.txt | cut -c1-200
rm -rf /tmp/vk; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/vk -o vk -- python "$OLDPWD/scripts/r6/vendor_kernels.py" > "$OLDPWD/$O/vendor_run.log" 2>&1)
python scripts/r6/vendor_kernels_parse.py "$(find /tmp/vk -name '*kernel_trace.csv' | head -1)" > $O/vendor_kernels.txt 2>/dev/null; grep -c . $O/vendor_kernels.txt
# steady-state kernel summaries: bf16 (side stream on), bf16 single stream, twin mode single stream
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
rm -rf /tmp/prof_out2
(cd /tmp && MAED_WGRAD_SIDE_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out2 -o bench -- python "$OLDPWD/bench.py" --steps 6 --warmup 3 --dtype f32 --f32-matmul bf16x3 --f32-backward bf16 --no-cpu-baseline > "$OLDPWD/$O/prof_twin.log" 2>&1)
tr2=$(find /tmp/prof_out2 -name "*kernel_trace.csv" | head -1)
python scripts/summarize_trace.py "$tr2" $O/rocprofv3_steady_state_kernels_twin_mode_single_stream.csv 4 | tail -1
# the default bench line once more WITH the PMC traffic of this very build in place (bench.py quotes traffic only for a matching source hash)
mkdir -p profiles/r06_pmc && cp $O/traffic.json profiles/r06_pmc/traffic.json
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_train_with_traffic.json 2> $O/bench_train_with_traffic.err; cut -c1-200 $O/bench_train_with_traffic.json
cp profiles/r06_pmc/traffic.json $O/traffic_stamped.json
