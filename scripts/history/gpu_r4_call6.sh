#!/bin/bash
# the contract's multi-GPU launch line with ONE rank and every bucket's all-reduce forced: library communicator (default) vs torch.distributed, with the per-stage
# weight standardisation a multi-rank job uses; plain single-process line beside them (same box)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4c6; mkdir -p $O; export TMPDIR=/tmp
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_plain.json 2>/dev/null; cut -c1-180 $O/bench_plain.json
for comm in direct torch; do
MAED_COMM=$comm MAED_FORCE_COLLECTIVES=1 MAED_WS_PER_STAGE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_world1_forced_$comm.json 2> $O/bench_world1_forced_$comm.err; cut -c1-180 $O/bench_world1_forced_$comm.json; grep -i "falling back\|error" $O/bench_world1_forced_$comm.err | head -3
done
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_plain2.json 2>/dev/null; cut -c1-180 $O/bench_plain2.json
python - <<'P'
import json
for f in ("plain","world1_forced_direct","world1_forced_torch","plain2"):
    try:
        d=json.load(open(f"gpurun_out/r4c6/bench_{f}.json")); print(f, d["ms_per_step"], (d.get("ddp") or {}).get("transport"))
    except Exception as e: print(f, "ERR", e)
P
