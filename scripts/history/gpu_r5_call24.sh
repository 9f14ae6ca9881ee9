#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5c24; rm -rf $O; mkdir -p $O
timeout 600 python scripts/x3p_forward_noise.py 2>&1 | grep -v amdgpu | tee $O/noise.txt
