#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3c9; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python scripts/x3_micro.py 30 tn all bf16 2>&1 | grep "^tn" | cut -c1-120 | tee $O/tn_after_lane_remap.txt
timeout 300 python scripts/x3_micro.py 20 conv all bf16 2>&1 | grep "^conv" | cut -c1-160 | tee -a $O/tn_after_lane_remap.txt
bash scripts/gpu_sq_x3.sh tn qkv bf16 gemm_tn_mfma 2>&1 | grep "BANK_CONFLICT\|IDX_ACTIVE\|MFMA_BUSY\|GUI_ACTIVE\|WAIT_ANY\|WAVE_CYCLES" | tee $O/sq_tn_after.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["step_time"]["median_ms"], "host", d.get("host_enqueue_ms"), json.dumps(d["roofline"])[:400])
PY
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "tn or wgrad or conv" --timeout=600 -p no:cacheprovider 2>&1 | tail -2
