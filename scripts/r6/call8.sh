#!/bin/bash
# round 6, call 8: new kernel-level GPU parity tests (planes / twins / persistent GEMM / deferred GroupNorm sums) + the attention residency measurement
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); O=gpurun_out/r6c8; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_planes_twins.py tests/test_gpu_kernels.py -m gpu -q --timeout=900 -p no:cacheprovider -k "planes or twin or persistent or deferred or split" > $O/pytest_new.log 2>&1; echo "pytest exit: $?" >> $O/pytest_new.log
grep -E "passed|failed|Error|error|assert" $O/pytest_new.log | tail -n 12
rm -rf /tmp/ar1 /tmp/ar2
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ar1 -o t -- python "$ROOT/scripts/r6/attn_residency.py" 8 > "$ROOT/$O/ar_trace.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/ar2 -o t -- python "$ROOT/scripts/r6/attn_residency.py" 4 > "$ROOT/$O/ar_pmc.log" 2>&1)
t1=$(find /tmp/ar1 -name "*kernel_trace.csv" | head -1); t2=$(find /tmp/ar2 -name "*kernel_trace.csv" | head -1); c2=$(find /tmp/ar2 -name "*counter_collection.csv" | head -1)
python scripts/r6/attn_residency_parse.py "$t1" > $O/attn_residency_durations.txt 2> $O/parse1.err; cat $O/attn_residency_durations.txt | cut -c1-200
python scripts/r6/attn_residency_parse.py "$t2" "$c2" > $O/attn_residency_fetch.txt 2> $O/parse2.err; cat $O/attn_residency_fetch.txt | cut -c1-220
head -2 "$c2" > $O/counter_header.txt
