#!/bin/bash
# same-box A/B of attention-forward build variants through MAED_HIP_LIB
cd "$(dirname "$0")/.."
for v in base cond noslp base cond noslp; do
  echo -n "$v: "; MAED_HIP_LIB=$PWD/ab_libs/lib_$v.so timeout 200 python scripts/attn_micro.py 100 2>&1 | grep attn_sp | cut -c1-60
done | tee gpurun_out/ab_attn.log
MAED_HIP_LIB=$PWD/ab_libs/lib_noslp.so timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "attn_spatial" 2>&1 | grep -E "passed|failed"
