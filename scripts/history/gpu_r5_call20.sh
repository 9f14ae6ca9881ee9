#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5c20; rm -rf $O; mkdir -p $O
timeout 600 python scripts/x3p_micro.py 20 3 2>&1 | tee $O/x3p_micro.txt
