#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5c11; rm -rf $O; mkdir -p $O
timeout 600 python scripts/tn_split_sweep_r5.py 2>&1 | tail -12 | tee $O/tn_split_sweep.txt
