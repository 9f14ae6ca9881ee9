#!/bin/bash
# round 6, second session, call 1: two wave groups per workgroup in the weight-gradient kernel -- micro-benchmark with split sweep, then the train step A/B (3 interleaved repeats)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r6b; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python scripts/r6b/tn_two_groups_micro.py 20 3 > $O/tn_two_groups_micro.txt 2>&1; tail -n 20 $O/tn_two_groups_micro.txt
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "wgrad or tn" -p no:cacheprovider 2>&1 | tail -n 3
for r in 1 2 3; do
  for v in 1 5; do
    MAED_TN_DMA=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ddp-rehearsal > $O/bench_tn${v}_$r.json 2> $O/bench_tn${v}_$r.err
    python - <<PY
import json
j = json.loads(open("$O/bench_tn${v}_$r.json").read().strip().splitlines()[-1])
print("MAED_TN_DMA=$v run $r:", j["ms_per_step"], "ms", j["value"], "clips/s; host enqueue", j.get("host_enqueue_ms"), "roofline", j["roofline"]["frac"], j["roofline"]["avg_us"])
PY
  done
done
