#!/bin/bash
# copies what scripts/gpu_final.sh left under gpurun_out/final/ (merged back by gpurun) into profiles/rNN_*:  bash scripts/collect_evidence.sh r02
set -eu
cd "$(dirname "$0")/.."
R=${1:?round tag, e.g. r02}; F=gpurun_out/final
for f in bench_train bench_forward bench_train_f32 bench_cfg5 bench_torchrun_world1_forced_collectives; do cp $F/$f.json profiles/${R}_$f.json; done
cp $F/gemm_micro.txt profiles/${R}_gemm_micro.txt; cp $F/gemm_tile_variants.txt profiles/${R}_gemm_tile_variants.txt
cp $F/parity_report_gpu.txt profiles/${R}_parity_report_gpu.txt; cp $F/pytest_gpu.log profiles/${R}_pytest_gpu.log; cp $F/smoke.log profiles/${R}_smoke.log
cp $F/rocprofv3_steady_state_kernels.csv profiles/${R}_rocprofv3_steady_state_kernels.csv
cp $F/rocprofv3_steady_state_kernels_single_stream.csv profiles/${R}_rocprofv3_steady_state_kernels_single_stream.csv 2>/dev/null || true
cp $F/rocprofv3_kernel_stats_incl_warmup.csv profiles/${R}_rocprofv3_kernel_stats_incl_warmup.csv
cp $F/rocprofv3_steady_state_kernels_last_step_sequence.txt profiles/${R}_rocprofv3_last_step_kernel_sequence.txt
mkdir -p profiles/${R}_pmc; cp $F/traffic.json profiles/${R}_pmc/traffic.json; cp gpurun_out/pmc/*.csv profiles/${R}_pmc/
python - <<PY
import json
from maed_amd.build import source_hash
t = json.load(open("profiles/${R}_pmc/traffic.json"))
print("traffic.json source hash", t["source_hash"], "== build", source_hash(), t["source_hash"] == source_hash())
PY
