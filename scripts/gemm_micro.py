"""bf16 MFMA GEMM at the cfg3 STE shapes in isolation (event timing; target for rocprofv3 --pmc).
usage: gemm_micro.py [iters] [shape|all] [impl,impl,...]   impl 2 = register-staged, 3/4 = direct global->LDS with 1/2 buffers"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd import ops, _lib as L
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
which = sys.argv[2] if len(sys.argv) > 2 else "all"
impls = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2]
torch.manual_seed(0)
M = 128 * 197
shapes = {"qkv": (M, 1536, 512, L.EPI_STORE), "fc1": (M, 2048, 512, L.EPI_GELU), "fc2": (M, 512, 2048, L.EPI_RESID_F32), "proj": (M, 512, 512, L.EPI_RESID_F32),
          "dfc2": (M, 2048, 512, L.EPI_MUL_DGELU), "dqkv": (M, 512, 1536, L.EPI_STORE), "dfc1": (M, 512, 2048, L.EPI_STORE), "dproj": (M, 512, 512, L.EPI_STORE),
          "sq4k": (4096, 4096, 4096, L.EPI_STORE),
          # the backbone's 1x1 convolutions at cfg3 (128 frames): stage 3 / 2 / 1, conv1- and conv3-shaped ("conv" selector)
          "c3a": (25088, 256, 1024, L.EPI_STORE), "c3b": (25088, 1024, 256, L.EPI_STORE), "c2a": (100352, 128, 512, L.EPI_STORE), "c2b": (100352, 512, 128, L.EPI_STORE),
          "c1a": (401408, 64, 256, L.EPI_STORE), "c1b": (401408, 256, 64, L.EPI_STORE)}
CONV = ("c3a", "c3b", "c2a", "c2b", "c1a", "c1b")
for name, (m, n, k, epi) in shapes.items():
    if name in CONV and which not in ("conv", name):
        continue
    if which == "conv" and name not in CONV:
        continue
    if which not in ("all", "conv", name) and not (which == "ste" and name != "sq4k"):      # "ste" = the eight NT GEMMs of one STE block (fwd + input gradients)
        continue
    A = [torch.randn(m, k, device="cuda").bfloat16() for _ in range(3)]
    B = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
    bias = torch.randn(n, device="cuda") if epi != L.EPI_MUL_DGELU else None
    aux = torch.randn(m, n, device="cuda") if epi == L.EPI_RESID_F32 else torch.randn(m, n, device="cuda").bfloat16() if epi == L.EPI_MUL_DGELU else None
    f32out = epi in (L.EPI_RESID_F32, L.EPI_STORE_F32)
    ref = None
    for impl in impls:
        out = torch.empty(m, n, device="cuda", dtype=torch.float32 if f32out else torch.bfloat16)
        out2 = torch.empty_like(out) if epi == L.EPI_GELU else None
        for a in A:
            ops.gemm_nt(a, B, epi, bias=bias, out=out, out2=out2, aux=aux, impl=impl)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            ops.gemm_nt(A[i % 3], B, epi, bias=bias, out=out, out2=out2, aux=aux, impl=impl)
        e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / iters
        ops.gemm_nt(A[0], B, epi, bias=bias, out=out, out2=out2, aux=aux, impl=impl)
        torch.cuda.synchronize()
        if ref is None:
            ref, diff = out.float().clone(), 0.0
        else:
            diff = (out.float() - ref).abs().max().item()
        print(f"gemm {name:5s} {m}x{n}x{k} impl {impl}: {us:8.2f} us  {2.0 * m * n * k / us / 1e6:7.1f} TFLOP/s ({2.0 * m * n * k / us / 1e6 / 2500:.3f} of MFMA peak)  max|diff vs first impl| = {diff:.3e}", flush=True)
