#!/bin/bash
# first measurement of the stem kernels: parity test, micro-benchmark, bench A/B (vendor stem vs own stem)
cd "$(dirname "$0")/.."
O=gpurun_out/stem1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "stem7x7" -p no:cacheprovider 2>&1 | tail -5
timeout 300 python scripts/stem_micro.py 20 > $O/stem_micro.txt 2>&1; cat $O/stem_micro.txt | grep -v amdgpu.ids
for own in 0 1; do
MAED_STEM_OWN=$own timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MAED_STEM_OWN=$own', d['ms_per_step'], d['value'])"
done
MAED_STEM_OWN=1 MAED_STEM_WREG=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('own+wreg', d['ms_per_step'], d['value'])"
