#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r02
{ timeout 200 python scripts/gemm_micro.py 30 all 0,6 2>&1 | grep "gemm "
  for nt in 2 3 4 6 16; do MAED_SWEEP_NT=$nt timeout 200 python scripts/gemm_micro.py 30 all 0,7 2>&1 | grep "impl 7" | sed "s/^/sweep_nt=$nt /"; done; } | tee gpurun_out/r02/gemm_variants_v2.txt
