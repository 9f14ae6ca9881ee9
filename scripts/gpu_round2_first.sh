#!/bin/bash
# First GPU call of round 2: everything added after the round-1 GPU budget ran out was verified on the host simulator only
# (tests/test_hostsim_{modes,iterative,eval,attention}.py).  This repeats those comparisons on the real library and times the
# new attention kernels.   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash scripts/gpu_round2_first.sh'
set -x
mkdir -p gpurun_out
timeout 900 python scripts/check_new_paths.py 2>&1 | tee gpurun_out/r02_new_paths_on_gpu.log
timeout 300 python scripts/attn_long_micro.py 20 2>&1 | tee gpurun_out/r02_attn_long_micro.txt
