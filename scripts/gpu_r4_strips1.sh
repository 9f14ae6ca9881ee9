#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "strips_of_rows or row_items" -p no:cacheprovider 2>&1 | tail -3
for v in 0 1; do echo "MAED_CONV3X3_WGRAD_STRIPS=$v"; MAED_CONV3X3_WGRAD_STRIPS=$v timeout 300 python scripts/conv3x3_micro.py 20 2>&1 | grep -E "stride 1|totals" | sed 's/.*| wgrad/wgrad/'; done
for wpb in 8 32; do echo "wpb=$wpb"; MAED_CONV3X3_STRIPS_WPB=$wpb timeout 300 python scripts/conv3x3_micro.py 20 2>&1 | grep -E "stride 1" | sed 's/.*| wgrad/wgrad/'; done
bash scripts/gpu_ab_env.sh MAED_CONV3X3_WGRAD_STRIPS 0 1 2
