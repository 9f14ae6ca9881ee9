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


@pytest.mark.parametrize("path", [os.path.join(ROOT, "profiles", "r01_bench_v10_train_27.6ms.json"), os.path.join(ROOT, "profiles", "r01_bench_v9_train_28.1ms.json")])
def test_committed_bench_lines_carry_the_contract_fields(path):
    d = json.load(open(path))
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
