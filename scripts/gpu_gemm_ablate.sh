#!/bin/bash
# where does the NT GEMM's time go?  ablation build (maed_amd/libmaed_hip_ablate.so: -DMAED_GEMM_ABLATE): 1 = no global stores, 2 = no loads, 4 = no MFMA
# usage: gpu_gemm_ablate.sh [impl]     (0 = 128x128 kernel, 6 = 256x256 pipelined kernel)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r02
IMPL=${1:-0}
export MAED_HIP_LIB=$PWD/maed_amd/libmaed_hip_ablate.so
for a in 0 1 2 4 3 5 6; do
  MAED_GEMM_ABLATE=$a timeout 120 python scripts/gemm_micro.py 30 all $IMPL 2>/dev/null | grep "gemm " | sed "s/^/ablate=$a /"
done | tee gpurun_out/r02/gemm_ablate_impl$IMPL.txt
