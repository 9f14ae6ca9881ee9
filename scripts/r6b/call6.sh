#!/bin/bash
# round 6, second session, call 6: one executable graph against two replayed in turn (MAED_GRAPHS)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r6b; mkdir -p $O; export TMPDIR=/tmp
for r in 1 2; do for n in 1 2 3; do
  MAED_GRAPHS=$n timeout 400 python bench.py --graph-leg --steps 20 --warmup 3 > $O/graph_leg_n${n}_$r.json 2> $O/graph_leg_n${n}_$r.err
  python - <<PY
import json
j = json.loads(open("$O/graph_leg_n${n}_$r.json").read().strip().splitlines()[-1])["graph"]
print("MAED_GRAPHS=$n run $r: replay", j["ms_per_step"], "ms  eager", j["eager_same_entry_points_ms_per_step"], " host per step back to back", j["host_ms_per_step"], " one replay idle", j["host_ms_one_replay_idle_queue"], " cpu", j["process_cpu_ms_per_step"], " max rel loss diff", "%.1e" % j["max_rel_loss_diff"])
PY
done; done
timeout 300 python -m pytest tests/test_gpu_graph.py -m gpu -q -p no:cacheprovider 2>&1 | tail -n 2
