#!/bin/bash
# copies what scripts/gpu_final_r6.sh left under gpurun_out/final/ (merged back by gpurun) into profiles/r06_*
set -eu
cd "$(dirname "$0")/.."
R=r06; F=gpurun_out/final
cp $F/bench_train_with_traffic.json profiles/${R}_bench_train.json
for f in bench_forward bench_train_f32_bf16x3 bench_train_f32_bf16x3_fwd_bf16x1_bwd bench_train_f32_bf16x3_fwd_bf16_twin_bwd bench_cfg5 bench_cfg5_twin; do cp $F/$f.json profiles/${R}_$f.json; done
for f in gemm_micro gemm_vs_vendor tn_micro attn_long_micro sk_micro tn_sk_micro vendor_kernels; do cp $F/$f.txt profiles/${R}_$f.txt; done
cp $F/graph_leg.json profiles/${R}_graph_leg.json; cp $F/test_gpu_graph.log profiles/${R}_test_gpu_graph.log; cp $F/stress_r6_kernels.txt profiles/${R}_stress_kernels.txt 2>/dev/null || true
cp $F/parity_report_gpu.txt profiles/${R}_parity_report_gpu.txt; cp $F/pytest_gpu.log profiles/${R}_pytest_gpu.log; cp $F/smoke.log profiles/${R}_smoke.log
cp $F/rocprofv3_steady_state_kernels.csv profiles/${R}_rocprofv3_steady_state_kernels.csv
cp $F/rocprofv3_steady_state_kernels_single_stream.csv profiles/${R}_rocprofv3_steady_state_kernels_single_stream.csv
cp $F/rocprofv3_steady_state_kernels_twin_mode_single_stream.csv profiles/${R}_rocprofv3_steady_state_kernels_twin_mode_single_stream.csv
cp $F/rocprofv3_kernel_stats_incl_warmup.csv profiles/${R}_rocprofv3_kernel_stats_incl_warmup.csv
cp $F/rocprofv3_steady_state_kernels_single_stream_last_step_sequence.txt profiles/${R}_rocprofv3_last_step_kernel_sequence.txt 2>/dev/null || true
mkdir -p profiles/${R}_pmc; cp $F/traffic_stamped.json profiles/${R}_pmc/traffic.json; cp gpurun_out/pmc/*.csv profiles/${R}_pmc/ 2>/dev/null || true
python - <<PY
import json
from maed_amd.build import source_hash
t = json.load(open("profiles/r06_pmc/traffic.json"))
print("traffic.json source hash", t["source_hash"], "== build", source_hash(), t["source_hash"] == source_hash())
PY
