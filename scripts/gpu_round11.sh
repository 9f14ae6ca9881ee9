#!/bin/bash
# pipelined spatial-attention forward (2 / 4 items per workgroup): parity under each setting + isolated timing
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
for it in 2 4; do
  MAED_ATTN_ITEMS=$it timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "attn_spatial" 2>&1 | grep -E "passed|failed" | sed "s/^/items=$it: /"
done
for it in 1 2 4; do echo "items=$it"; MAED_ATTN_ITEMS=$it timeout 200 python scripts/attn_micro.py 50 2>&1 | grep attn_sp; done | tee gpurun_out/attn_items.log
