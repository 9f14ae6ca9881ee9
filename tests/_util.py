"""shared helpers of the GPU parity tests"""
import os

import torch

DEV = "cuda"
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.txt")


def report(name, got, ref, rtol, atol):
    """log max errors to gpurun_out/parity_report.txt, then assert |got-ref| <= atol + rtol*|ref| elementwise"""
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    err = (got - ref).abs()
    denom = ref.abs().max().item() + 1e-30
    line = f"{name:58s} max_abs={err.max().item():.3e} rel_to_max={err.max().item() / denom:.3e} ref_max={denom:.3e} nan={int(torch.isnan(got).sum())}"
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as fh:
        fh.write(line + "\n")
    print(line)
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert not torch.isnan(got).any(), name
    bad = err > (atol + rtol * ref.abs())
    assert not bad.any(), f"{name}: {int(bad.sum())}/{bad.numel()} elements out of tolerance; {line}"


def note(line):
    """free-form line into the parity report"""
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as fh:
        fh.write(line + "\n")
    print(line)


def tol(dtype, scale=1.0):
    return (dict(rtol=2e-5, atol=2e-5 * scale) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2 * scale))


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def q(x, dtype):
    """round-trip through the compute dtype (what the kernel actually sees)"""
    return x.to(dtype).float()
