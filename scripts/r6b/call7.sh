#!/bin/bash
# round 6, second session, call 7: 3x3 convolutions on one frame per workgroup -- micro-benchmark, conv GPU tests, train step A/B (3 interleaved repeats)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r6b; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python scripts/r6b/conv3x3_frame_micro.py 20 2>&1 | grep -v amdgpu.ids | tee $O/conv3x3_frame_micro.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "conv or backbone or resnet or cfg3 or train" -p no:cacheprovider 2>&1 | tail -n 3
for r in 1 2 3; do
  for v in 0 1; do
    MAED_CONV3X3_FRAME=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ddp-rehearsal > $O/bench_cf${v}_$r.json 2> $O/bench_cf${v}_$r.err
    python - <<PY
import json
j = json.loads(open("$O/bench_cf${v}_$r.json").read().strip().splitlines()[-1])
print("MAED_CONV3X3_FRAME=$v run $r:", j["ms_per_step"], "ms", j["value"], "clips/s; loss", j.get("first_step_loss"))
PY
  done
done
