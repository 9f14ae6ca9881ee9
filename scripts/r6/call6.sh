#!/bin/bash
# round 6, call 6: the whole GPU suite at HEAD + interleaved A/B of the bench: weight-gradient kernel by shape (MAED_TN_DMA=4) vs the LDS-DMA kernel everywhere (=1)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r6c6; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest exit: $?" >> $O/pytest_gpu.log
grep -E "passed|failed|Error|error" $O/pytest_gpu.log | tail -n 8
for r in 1 2 3; do
  for v in 1 4; do
    MAED_TN_DMA=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ddp-rehearsal > $O/bench_tn${v}_$r.json 2> $O/bench_tn${v}_$r.err
    python - <<PY
import json
j = json.loads(open("$O/bench_tn${v}_$r.json").read().strip().splitlines()[-1])
print("MAED_TN_DMA=$v run $r:", j["ms_per_step"], "ms", j["value"], "clips/s; host enqueue", j.get("host_enqueue_ms"))
PY
  done
done
