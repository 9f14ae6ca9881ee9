"""The bench.py contract, checked without a GPU: the CLI the driver calls exists, and the JSON lines this tree produced on hardware
(committed under profiles/) carry every field the contract names, with the metric / unit BASELINE.json prescribes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOP = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
       "roofline", "cpu_baseline"}
ROOFLINE = {"bound", "achieved", "peak", "unit", "frac", "traffic"}
CPU = {"value", "unit", "cores", "kind", "sample"}


def test_cli_flags_of_the_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


@pytest.mark.parametrize("path", [os.path.join(ROOT, "profiles", "r02_bench_train.json"), os.path.join(ROOT, "profiles", "r02_bench_torchrun_world1_forced_collectives.json"),
                                  os.path.join(ROOT, "profiles", "r01_bench_v10_train_27.6ms.json")])
def test_committed_bench_lines_carry_the_contract_fields(path):
    d = json.loads(open(path).read().strip().split("\n")[-1])
    assert TOP <= set(d), TOP - set(d)
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["unit"] == "video-clips/sec" and base["metric"].startswith(d["unit"]) and d["metric"].startswith(d["unit"])
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and d["steps"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] - 8 * d["n_gpus"] * 1e3 / d["ms_per_step"]) < 0.02 * d["value"]       # whole-job clips/s = clips per step / step time
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert ROOFLINE <= set(r) and r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    if d["cpu_baseline"] is not None:                                                         # (absent in --no-cpu-baseline A/B runs)
        c = d["cpu_baseline"]
        assert CPU <= set(c) and c["kind"] in ("port", "reference") and c["cores"] >= 1
    if "step_time" in d:                                                                      # round 2 on: per-step hipEvent statistics
        st = d["step_time"]
        assert st["p10_ms"] <= st["median_ms"] <= st["p90_ms"] and abs(st["median_ms"] - d["ms_per_step"]) < 0.1 * d["ms_per_step"]


def test_committed_pmc_traffic_is_stamped_with_a_source_hash():
    """bench.py quotes roofline.traffic only from a PMC file measured on the running build: the committed file names the hash of the kernel sources it
    was collected on and the per-launch bytes of the three roofline kernels"""
    t = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc", "traffic.json")))
    assert len(t["source_hash"]) == 16 and set(t["kernels"]) == {"attn_spatial_fwd", "gemm_nt", "gemm_tn"}
    assert all(isinstance(v, int) and v > 0 for v in t["kernels"].values())
    # the attention forward moves its algorithmic bytes and nothing more (4*M*C*2 + lse at cfg3 = 103.6 MB)
    assert abs(t["kernels"]["attn_spatial_fwd"] - 103.6e6) < 0.02 * 103.6e6
