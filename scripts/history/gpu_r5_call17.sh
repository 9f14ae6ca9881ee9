#!/bin/bash
# A/B of two library builds on the same box: maed_amd/libmaed_hip_base.so (HEAD) vs maed_amd/libmaed_hip.so (candidate)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5c17; rm -rf $O; mkdir -p $O
for v in base cand base cand base cand; do
lib=maed_amd/libmaed_hip.so; [ $v = base ] && lib=maed_amd/libmaed_hip_base.so
MAED_HIP_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$v', d['ms_per_step'], d['value'], 'fc1', k['gemm_fc1_gelu']['avg_us'], 'fc2', k['gemm_fc2_resid']['avg_us'])" | tee -a $O/ab.txt
done
