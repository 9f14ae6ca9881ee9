#!/bin/bash
# round 5, call 12: the LBS blend on the fp32 matrix cores vs the VALU kernel (MAED_LBS_FB=16) -- micro, tail tests, forward-only and train-step A/B
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5c12; rm -rf $O; mkdir -p $O
for fb in 16 0 16 0; do echo "MAED_LBS_FB=$fb: $(MAED_LBS_FB=$fb timeout 120 python scripts/lbs_micro.py 50 2>/dev/null | head -1)" | tee -a $O/lbs_micro.txt; done
timeout 900 python -m pytest tests/test_gpu_tail.py tests/test_gpu_kernels.py tests/test_gpu_model.py -q -x -k "tail or smpl or lbs or cfg1 or cfg2" -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest.log
for fb in 16 0 16 0; do
MAED_LBS_FB=$fb timeout 300 python bench.py --steps 20 --warmup 5 --forward-only --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('forward MAED_LBS_FB=$fb', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
done
for fb in 16 0 16 0; do
MAED_LBS_FB=$fb timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train MAED_LBS_FB=$fb', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
done
