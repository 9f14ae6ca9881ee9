#!/bin/bash
# First-contact GPU run: every test group in its own process (a GPU fault in one kernel must not hide the
# others), then smoke + a short bench.  Logs under gpurun_out/.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt gpurun_out/dbg_*.log
export TMPDIR=/tmp
groups=("test_gpu_kernels.py -k layernorm" "test_gpu_kernels.py -k gemm" "test_gpu_kernels.py -k transpose"
        "test_gpu_kernels.py -k attn_spatial_fwd" "test_gpu_kernels.py -k attn_spatial_bwd" "test_gpu_kernels.py -k attn_temporal"
        "test_gpu_kernels.py -k st_mix" "test_gpu_kernels.py -k embed" "test_gpu_kernels.py -k adam" "test_gpu_kernels.py -k ktd"
        "test_gpu_kernels.py -k rot6d" "test_gpu_kernels.py -k smpl"
        "test_gpu_model.py -k golden_f32" "test_gpu_model.py -k block_forward_backward" "test_gpu_model.py -k maed_forward_small"
        "test_gpu_model.py -k cfg1" "test_gpu_model.py -k gradients" "test_gpu_model.py -k train_step")
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  log="gpurun_out/dbg_$(printf %02d $i).log"
  echo "### tests/$g" > "$log"
  timeout 600 python -m pytest tests/$g -m gpu -q --timeout=300 -p no:cacheprovider -x >> "$log" 2>&1
  echo "exit=$?" >> "$log"
  echo "[$i] $g -> $(tail -n 2 "$log" | tr '\n' ' ')"
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?" >> gpurun_out/smoke.log; tail -n 3 gpurun_out/smoke.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench exit: $?" >> gpurun_out/bench.log; tail -n 3 gpurun_out/bench.log
