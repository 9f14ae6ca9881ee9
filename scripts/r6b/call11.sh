#!/bin/bash
# round 6, second session, call 11: maxpool backward on 2 x 2 input blocks -- parity test, kernel time under rocprofv3, bench
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r6b; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "maxpool" -p no:cacheprovider 2>&1 | tail -n 2
rm -rf /tmp/mp; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mp -o mp -- python "$OLDPWD/bench.py" --steps 6 --warmup 3 --no-cpu-baseline --no-ddp-rehearsal > /dev/null 2>&1)
grep -h "maxpool" $(find /tmp/mp -name "*kernel_stats.csv" | head -1) | cut -c1-140
for r in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ddp-rehearsal 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'])"; done
