#!/bin/bash
# where does the plane kernel's time go?  ablation build: 1 = no stores, 2 = no copies after the prologue, 4 = no MFMA / fragment reads
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5c21; rm -rf $O; mkdir -p $O
export MAED_HIP_LIB=$PWD/maed_amd/libmaed_hip_ablate.so
for a in 0 1 2 4 3 5 6 7; do
  MAED_GEMM_ABLATE=$a timeout 200 python scripts/x3p_micro.py 20 2 2>/dev/null | grep "^nt " | sed "s/^/ablate=$a /" | cut -c1-330
done | tee $O/x3p_ablate.txt
