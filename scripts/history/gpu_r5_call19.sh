#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5c19; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_mode.py -q -x -p no:cacheprovider -k "train_steps_track" 2>&1 | tail -8 | tee $O/pytest.log
grep "six train steps" gpurun_out/parity_report.txt | tee $O/curves.txt
