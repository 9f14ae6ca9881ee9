#!/bin/bash
# First GPU call of round 2 (~18 GPU-minutes): everything added after the round-1 GPU budget ran out was verified on the host simulator
# only.  This (1) repeats those parity checks on the real library, (2) times the new attention kernels against the measured ones,
# (3) A/Bs the opt-in switches (all written from ISA / trace analysis without a GPU) in one bench run against the default, and
# (4) runs the GPU suite with the switches on.   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash scripts/gpu_round2_first.sh'
set -x
mkdir -p gpurun_out
OPTIN="MAED_GN_DEFER_AFFINE=1 MAED_LN_DEFER_AFFINE=1 MAED_TAIL_PARALLEL=1 MAED_TM_BWD_L32=1 MAED_TM_BWD_WIDE_REGS=1 MAED_WS_PER_STAGE=1"
bench_ms() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d['ms_per_step'], 'ms/step', {k: v['avg_us'] for k, v in d['kernels'].items()})" "$1"; }

MAED_RUN_UNVERIFIED_GPU_TESTS=1 timeout 600 python -m pytest tests/test_gpu_unverified.py -m gpu -q -s 2>&1 | tee gpurun_out/r02_new_paths_on_gpu.log
timeout 200 python scripts/attn_long_micro.py 20 2>&1 | tee gpurun_out/r02_attn_long_micro.txt
timeout 200 python scripts/conv3x3_micro.py 10 2>&1 | tee gpurun_out/r02_conv3x3_micro.txt
for w in 0 1; do MAED_TM_BWD_WIDE_REGS=$w MAED_TEMPORAL_MFMA=1 timeout 120 python scripts/attn_tm_micro.py 30 2>&1 | sed "s/^/wide_regs=$w /" | tee -a gpurun_out/r02_attn_tm_wide_regs.txt; done
MAED_TM_BWD_L32=1 MAED_TEMPORAL_MFMA=1 timeout 120 python scripts/attn_tm_micro.py 30 2>&1 | sed "s/^/one_tile_kernel /" | tee -a gpurun_out/r02_attn_tm_wide_regs.txt

timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | bench_ms "default" | tee gpurun_out/r02_optin_flags.txt
env $OPTIN timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | bench_ms "all-opt-in" | tee -a gpurun_out/r02_optin_flags.txt
env MAED_CONV3X3=own timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | bench_ms "MAED_CONV3X3=own" | tee -a gpurun_out/r02_optin_flags.txt
env MAED_GN_DEFER_AFFINE=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | bench_ms "MAED_GN_DEFER_AFFINE" | tee -a gpurun_out/r02_optin_flags.txt
# (LayerNorm deferral, chain kernels, per-stage standardisation: read their effect off the per-kernel averages of the two runs above /
#  a rocprofv3 pass; a bench run each costs ~1.5 GPU-minutes)
env $OPTIN timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee -a gpurun_out/r02_optin_flags.txt
