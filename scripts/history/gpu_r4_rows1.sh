#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "row_items" -p no:cacheprovider 2>&1 | tail -3
for wg in 0 128 192 256; do
if [ $wg = 0 ]; then export MAED_CONV3X3_WGRAD_ROWS=0; else export MAED_CONV3X3_WGRAD_ROWS=1 MAED_CONV3X3_ROWS_WGS=$wg; fi
echo "rows=$MAED_CONV3X3_WGRAD_ROWS wgs=$wg: $(timeout 300 python scripts/conv3x3_micro.py 20 2>&1 | grep 'H= 56 C=  64' | sed 's/.*| wgrad/wgrad/')"
done
unset MAED_CONV3X3_ROWS_WGS
bash scripts/gpu_ab_env.sh MAED_CONV3X3_WGRAD_ROWS 0 1 2
