#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4c3; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python scripts/gn_bwd_micro.py 20 > $O/gn_bwd_micro.txt 2>&1; cat $O/gn_bwd_micro.txt | tail -16
