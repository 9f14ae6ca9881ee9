#!/bin/bash
# NT GEMM tile variants at the cfg3 STE shapes, same box: impl 3 = 128x128 one LDS buffer (4 workgroups per CU), 4 = 128x128 two buffers,
# 6 = 256x256 pipelined (gemm256.hip), 0 = what the dispatch rule picks -> gpurun_out/final/gemm_tile_variants.txt
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/final
timeout 300 python scripts/gemm_micro.py 30 all 3,4,6,0 2>&1 | grep "gemm " | tee gpurun_out/final/gemm_tile_variants.txt
