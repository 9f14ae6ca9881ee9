#!/bin/bash
# round 6, call 5: the whole GPU suite at HEAD (persistent K-stream GEMM, deferred GroupNorm affine gradients, ADVICE r5 fixes) + interleaved A/B of the bench:
# GroupNorm closing sums per layer (MAED_GN_DEFER_AFFINE=0) vs one batched launch
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r6c5; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; echo "pytest exit: $?" >> $O/pytest_gpu.log
grep -E "passed|failed|Error|error" $O/pytest_gpu.log | tail -n 6
for r in 1 2 3; do
  for v in 0 1; do
    MAED_GN_DEFER_AFFINE=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ddp-rehearsal > $O/bench_defer${v}_$r.json 2> $O/bench_defer${v}_$r.err
    python - <<PY
import json
j = json.loads(open("$O/bench_defer${v}_$r.json").read().strip().splitlines()[-1])
print("defer=$v run $r:", j["ms_per_step"], "ms", j["value"], "clips/s; host enqueue", j.get("host_enqueue_ms"))
PY
  done
done
