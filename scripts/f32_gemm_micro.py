"""the decoder head's fp32 GEMMs (KTD fc1 / fc2 / packed regressor, forward and input gradients; F = 128 frames) on the exact-fp32 kernel"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd import ops, _lib as L
shapes = [("fc1", 128, 1024, 2048 + 85), ("fc2", 128, 1024, 1024), ("feat", 128, 157, 1024), ("dfc1", 128, 2133, 1024), ("dfeat", 128, 1024, 157)]
tot = 0.0
for name, M, N, K in shapes:
    K = (K + 3) // 4 * 4
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda") * K ** -0.5; bias = torch.randn(N, device="cuda")
    ref = (A.double() @ B.double().t() + bias.double())
    out = ops.gemm_nt(A, B, L.EPI_STORE, bias=bias)
    err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.gemm_nt(A, B, L.EPI_STORE, bias=bias)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 50; tot += us
    print(f"f32 gemm {name:5s} {M}x{N}x{K}: {us:6.1f} us  rel err {err:.1e}")
print(f"total {tot:.1f} us")
