#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r6b; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_graph.py -m gpu -q -s -p no:cacheprovider > $O/test_graph.log 2>&1; grep -v "amdgpu.ids" $O/test_graph.log | tail -n 40
timeout 400 python bench.py --graph-leg --steps 20 --warmup 3 > $O/graph_leg.json 2> $O/graph_leg.err; tail -n 12 $O/graph_leg.err; cat $O/graph_leg.json
