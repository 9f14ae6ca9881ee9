#!/bin/bash
# GEMM 1x1 convolutions vs all-MIOpen on the SAME box: phase timings, then the steady-state profile of the GEMM path
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 1 0; do
  echo "== MAED_GEMM_CONVS=$v"
  MAED_GEMM_CONVS=$v timeout 600 python scripts/diag_step.py 2>&1 | grep -E "backbone fwd|fwd\+bwd|train step" | grep -vE "#0" | cut -c1-90
done
bash scripts/gpu_prof.sh > gpurun_out/prof_stdout.log 2>&1; head -4 gpurun_out/prof_stdout.log | cut -c1-200
