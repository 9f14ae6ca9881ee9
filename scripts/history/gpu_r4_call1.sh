#!/bin/bash
# round 4, call 1: box baseline (short bench line), vendor GEMM yardstick, attention micro-benchmarks at the benchmarked shape
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4c1; mkdir -p $O; export TMPDIR=/tmp
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_train.json 2> $O/bench_train.err; cut -c1-220 $O/bench_train.json
timeout 300 python scripts/gemm_vs_vendor.py 30 > $O/gemm_vs_vendor.txt 2>&1; cat $O/gemm_vs_vendor.txt
TORCH_BLAS_PREFER_HIPBLASLT=0 timeout 300 python scripts/gemm_vs_vendor.py 30 > $O/gemm_vs_vendor_rocblas.txt 2>&1; grep -c . $O/gemm_vs_vendor_rocblas.txt
timeout 200 python scripts/attn_long_micro.py > $O/attn_long_micro.txt 2>&1; tail -12 $O/attn_long_micro.txt
