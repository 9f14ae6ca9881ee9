#!/bin/bash
# which MIOpen solver family costs what: the asm implicit-GEMM NHWC solvers zero-fill their output first (SubTensorOpWithScalar1d,
# 1.4 ms/step) and cast through an fp32 workspace (SubTensorOpWithCastTensor1d, 0.8 ms/step); try the fallbacks
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { echo "== $1"; env $1 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step', d['ms_per_step'])" || echo "   failed"; }
run "MAED_NOOP=1"
run "MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC=0"
run "MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC=0"
run "MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS_NHWC=0"
run "MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS_NHWC=0"
