#!/bin/bash
# where does the TN weight-gradient GEMM's time go?  ablation build: 1 = no atomics, 2 = no global loads, 4 = no MFMA / fragment reads,
# 8 = no transposing LDS stores (sums of those combine).  STE shapes + the backbone's 1x1 shapes (scripts/wgrad_micro.py)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r02
export MAED_HIP_LIB=$PWD/maed_amd/libmaed_hip_ablate.so
for a in 0 1 2 4 8 3 12 14 15; do
  MAED_GEMM_ABLATE=$a WGRAD_STE=1 timeout 200 python scripts/wgrad_micro.py 20 2>/dev/null | grep "gemm_tn\|STE wgrad" | sed "s/^/ablate=$a /"
done | tee gpurun_out/r02/tn_ablate.txt
