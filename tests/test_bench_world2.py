"""bench.py's own main() end to end with TWO ranks: process-group init -> parameter broadcast -> train steps with the bucketed gradient
all-reduce launched from the backward pass -> the extra profiling steps every rank must take part in -> barriers -> rank 0's JSON line.  No GPU here, so
`--simulate` swaps in CPU tensors, the host simulator and the gloo backend (tiny workload); what is under test is that the orchestration the driver's
N = 2/4/8 runs take cannot hang or desynchronise, and that both ranks end with identical parameters."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
def test_bench_main_two_ranks_gloo_simulator():
    port = 29700 + os.getpid() % 200
    env = dict(os.environ, MAED_SYNTHETIC_SMPL_OK="1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--simulate", "--dtype", "f32", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                      # rank 0 prints ONE JSON line
    # ... and nothing else reaches stdout from bench.py: native libraries' prints (RCCL's version banner at communicator creation) go to stderr
    assert [l for l in out.stdout.splitlines() if l.strip() and not l.startswith("{")] == [] or "torch.distributed" in out.stdout, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and abs(d["value"] - 2 * 1 * 1e3 / d["ms_per_step"]) < 0.05 * d["value"]      # whole-job clips/s: 2 ranks x 1 clip
    ddp = d["ddp"]
    assert ddp["rccl_ranks"] == 2 and ddp["collectives"] is True and ddp["transport"].startswith("torch.distributed/gloo")
    assert len(ddp["buckets"]) >= 1 and sorted(ddp["bucket_launch_order"]) == list(range(len(ddp["buckets"])))
    assert ddp["per_stage_weight_std"] is True
