#!/bin/bash
# round 6, call 2: first run of the persistent K-stream GEMM on the GPU: correctness + determinism + time per shape; vendor kernel names (fixed marker parse)
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); O=gpurun_out/r6c2; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python scripts/r6/sk_micro.py 20 20 > $O/sk_micro.txt 2> $O/sk_micro.err; echo "sk_micro exit $?" >> $O/sk_micro.txt
cat $O/sk_micro.txt | cut -c1-330; tail -5 $O/sk_micro.err
rm -rf /tmp/vk
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/vk -o vk -- python "$ROOT/scripts/r6/vendor_kernels.py" > "$ROOT/$O/vendor_run.log" 2>&1)
tr=$(find /tmp/vk -name "*kernel_trace.csv" | head -1)
python scripts/r6/vendor_kernels_parse.py "$tr" > $O/vendor_kernels.txt 2>$O/parse.err; cut -c1-200 $O/vendor_kernels.txt | head -60
