#!/bin/bash
# First GPU call of round 2: everything added after the round-1 GPU budget ran out was verified on the host simulator only
# (tests/test_hostsim_{modes,iterative,eval,attention}.py).  This repeats those comparisons on the real library and times the
# new attention kernels.   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash scripts/gpu_round2_first.sh'
set -x
mkdir -p gpurun_out
timeout 900 python scripts/check_new_paths.py 2>&1 | tee gpurun_out/r02_new_paths_on_gpu.log
timeout 300 python scripts/attn_long_micro.py 20 2>&1 | tee gpurun_out/r02_attn_long_micro.txt
# temporal backward: measured default (1024-thread register budget, 40 VGPRs spilled) vs the spill-free instantiation
for w in 0 1; do MAED_TM_BWD_WIDE_REGS=$w MAED_TEMPORAL_MFMA=1 timeout 120 python scripts/attn_tm_micro.py 30 2>&1 | sed "s/^/wide_regs=$w /" | tee -a gpurun_out/r02_attn_tm_wide_regs.txt; done
# decoder tail: thread-per-frame chain kernels (measured) vs the lane-parallel ones (bit-identical on the simulator)
for w in 0 1; do MAED_TAIL_PARALLEL=$w timeout 300 python -m pytest tests/test_gpu_tail.py -q -x 2>&1 | tail -2 | sed "s/^/tail_parallel=$w /" | tee -a gpurun_out/r02_tail_parallel.txt; done
for w in 0 1; do MAED_TAIL_PARALLEL=$w timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tail_parallel=$w', d['ms_per_step'], 'ms/step')" | tee -a gpurun_out/r02_tail_parallel.txt; done
# GroupNorm backward: per-workgroup dgamma/dbeta atomics (measured) vs deferred frame sums; then every opt-in together
for flags in "MAED_GN_DEFER_AFFINE=1" "MAED_LN_DEFER_AFFINE=1" "MAED_GN_DEFER_AFFINE=1 MAED_LN_DEFER_AFFINE=1 MAED_TAIL_PARALLEL=1 MAED_TM_BWD_WIDE_REGS=1"; do
  env $flags timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$flags', d['ms_per_step'], 'ms/step')" | tee -a gpurun_out/r02_optin_flags.txt
done
env MAED_GN_DEFER_AFFINE=1 MAED_LN_DEFER_AFFINE=1 MAED_TAIL_PARALLEL=1 MAED_TM_BWD_WIDE_REGS=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee -a gpurun_out/r02_optin_flags.txt
