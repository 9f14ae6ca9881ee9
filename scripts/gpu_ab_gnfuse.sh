set -u
mkdir -p gpurun_out/gnfuse
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu -x -k "statistics or backbone or cfg3 or groupnorm or conv" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
for f in 1 0 1 0; do MAED_GN_FUSE_STATS=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fuse=$f', d['ms_per_step'], d['value'])"; done
BENCH_EXTRA="" bash scripts/gpu_prof.sh > gpurun_out/gnfuse/prof.txt 2>&1; grep -E "st_colmean|st_mix_bwd|gn_stats|gn_apply|glds|conv3x3" gpurun_out/prof/steady_state_kernels.csv | head -30
