#!/bin/bash
# round 6, second session, call 4: parameter gradients written straight into .grad at the single-use call sites (MAED_DIRECT_GRADS=1, default) against autograd's way (=0):
# the GPU suite, then 3 interleaved repeats of the train step
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r6b; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $O/pytest_gpu_direct.log 2>&1; echo "pytest exit: $?" >> $O/pytest_gpu_direct.log
grep -E "passed|failed|Error" $O/pytest_gpu_direct.log | tail -n 6
for r in 1 2 3; do
  for v in 0 1; do
    MAED_DIRECT_GRADS=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ddp-rehearsal > $O/bench_dg${v}_$r.json 2> $O/bench_dg${v}_$r.err
    python - <<PY
import json
j = json.loads(open("$O/bench_dg${v}_$r.json").read().strip().splitlines()[-1])
print("MAED_DIRECT_GRADS=$v run $r:", j["ms_per_step"], "ms", j["value"], "clips/s; host enqueue", j.get("host_enqueue_ms"), "loss", j.get("first_step_loss"), "aten launches", (j.get("kernel_groups") or {}).get("groups", {}).get("aten_runtime"))
PY
  done
done
