#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4c15; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python scripts/gemm_vs_vendor.py 30 > $O/gemm_vs_vendor.txt 2>&1; cat $O/gemm_vs_vendor.txt | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_parity_mode.py -m gpu -x -q -k "mixed" -p no:cacheprovider > $O/pytest.log 2>&1; tail -6 $O/pytest.log
