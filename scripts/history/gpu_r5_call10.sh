#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5c10; rm -rf $O; mkdir -p $O
timeout 600 python scripts/stress_r5_kernels.py 200 2>&1 | tail -6 | tee $O/stress.txt
