#!/bin/bash
# round 6, second session, call 8: weight-gradient kernel by shape (MAED_TN_DMA=4: the register-transposing kernel for M >= 65536 rows onto <= 65536 outputs) against the
# LDS-DMA kernel everywhere (=1), at HEAD: 4 interleaved repeats, step time and the roofline line (single-stream hipEvents over every launch of the family)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r6b; mkdir -p $O; export TMPDIR=/tmp
for r in 1 2 3 4; do
  for v in 1 4; do
    MAED_TN_DMA=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ddp-rehearsal > $O/bench_tnd${v}_$r.json 2> $O/bench_tnd${v}_$r.err
    python - <<PY
import json
j = json.loads(open("$O/bench_tnd${v}_$r.json").read().strip().splitlines()[-1])
print("MAED_TN_DMA=$v run $r:", j["ms_per_step"], "ms", j["value"], "clips/s; roofline frac", j["roofline"]["frac"], "avg_us", j["roofline"]["avg_us"], "ms/step", j["roofline"]["ms_per_step"])
PY
  done
done
