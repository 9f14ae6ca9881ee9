#!/bin/bash
# round 3: DMA-staged bf16 weight-gradient kernel vs the register-staged one (micro, then the train step)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3c6; mkdir -p $O; export TMPDIR=/tmp
for k in 0 1; do echo "== MAED_TN_KERNEL=$k"; MAED_TN_KERNEL=$k timeout 300 python scripts/x3_micro.py 30 tn all bf16 2>&1 | grep "^tn" | cut -c1-120; done | tee $O/tn_kernel_ab.txt
for k in 0 1; do MAED_TN_KERNEL=$k timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_tn$k.json 2> $O/bench_tn$k.err; echo "bench TN_KERNEL=$k exit $?"; python - <<PY
import json
d=json.loads(open("$O/bench_tn$k.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["step_time"]["median_ms"], json.dumps(d["roofline_wgrad"])[:300])
PY
done
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_mode.py -m gpu -q -k "tn or wgrad or parity" --timeout=600 -p no:cacheprovider 2>&1 | tail -4
