#!/bin/bash
# round 6, second session, call 2: the STE blocks' weight-gradient GEMMs alternating between TWO library side streams (MAED_WGRAD_SIDE_STREAM=2) against one (=1): 3 interleaved repeats
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r6b; mkdir -p $O; export TMPDIR=/tmp
for r in 1 2 3; do
  for v in 1 2; do
    MAED_WGRAD_SIDE_STREAM=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ddp-rehearsal > $O/bench_ss${v}_$r.json 2> $O/bench_ss${v}_$r.err
    python - <<PY
import json
j = json.loads(open("$O/bench_ss${v}_$r.json").read().strip().splitlines()[-1])
print("MAED_WGRAD_SIDE_STREAM=$v run $r:", j["ms_per_step"], "ms", j["value"], "clips/s; host enqueue", j.get("host_enqueue_ms"), "loss", j.get("first_step_loss"))
PY
  done
done
