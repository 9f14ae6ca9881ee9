"""bf16 MFMA GEMM at the cfg3 STE shapes in isolation (event timing; target for rocprofv3 --pmc)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd import ops, _lib as L
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
which = sys.argv[2] if len(sys.argv) > 2 else "all"
torch.manual_seed(0)
M = 128 * 197
shapes = {"qkv": (M, 1536, 512, L.EPI_STORE), "fc1": (M, 2048, 512, L.EPI_GELU), "fc2": (M, 512, 2048, L.EPI_RESID_F32), "proj": (M, 512, 512, L.EPI_RESID_F32)}
for name, (m, n, k, epi) in shapes.items():
    if which not in ("all", name):
        continue
    A = [torch.randn(m, k, device="cuda").bfloat16() for _ in range(3)]
    B = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
    bias = torch.randn(n, device="cuda")
    aux = torch.randn(m, n, device="cuda") if epi == L.EPI_RESID_F32 else None
    out = torch.empty(m, n, device="cuda", dtype=torch.float32 if epi == L.EPI_RESID_F32 else torch.bfloat16)
    out2 = torch.empty_like(out) if epi == L.EPI_GELU else None
    for a in A:
        ops.gemm_nt(a, B, epi, bias=bias, out=out, out2=out2, aux=aux, impl=2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        ops.gemm_nt(A[i % 3], B, epi, bias=bias, out=out, out2=out2, aux=aux, impl=2)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters
    print(f"gemm {name:5s} {m}x{n}x{k}: {us:8.2f} us  {2.0 * m * n * k / us / 1e6:7.1f} TFLOP/s ({2.0 * m * n * k / us / 1e6 / 2500:.3f} of MFMA peak)")
