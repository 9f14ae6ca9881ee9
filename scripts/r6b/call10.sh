#!/bin/bash
# round 6, second session, call 10: one-pass GroupNorm backward with 256-thread workgroups (MAED_GN_BWD_ONEPASS=2) against 512 (=1) at HEAD, 3 interleaved repeats
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r6b; mkdir -p $O; export TMPDIR=/tmp
for r in 1 2 3; do for v in 1 2; do
  MAED_GN_BWD_ONEPASS=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ddp-rehearsal 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MAED_GN_BWD_ONEPASS=$v run $r:', j['ms_per_step'], j['value'])"
done; done
