"""STE temporal attention forward/backward in isolation (bf16, the MFMA kernels of csrc/attn_temporal.hip) at the cfg3 and cfg5 shapes.
(The round-1 MAED_TEMPORAL_MFMA=0/1 A/B switch is gone with the LDS-staged bf16 kernels it selected: this script times ONE implementation.)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd import ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for name, (F_, P, H, T) in {"cfg3": (128, 197, 8, 16), "cfg5": (128, 257, 12, 64)}.items():
    C = 64 * H
    torch.manual_seed(0)
    qkv = [torch.randn(F_, P, 3 * C, device="cuda").bfloat16() for _ in range(3)]
    do = torch.randn(F_, P, C, device="cuda").bfloat16()
    o, lse = ops.attn_temporal_fwd(qkv[0], H, T)
    dq = ops.attn_temporal_bwd(qkv[0], o, do, lse, H, T)
    torch.cuda.synchronize()
    res = {}
    for tag, fn in (("fwd", lambda i: ops.attn_temporal_fwd(qkv[i % 3], H, T)), ("bwd", lambda i: ops.attn_temporal_bwd(qkv[i % 3], o, do, lse, H, T, dqkv=dq))):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(i)
        e1.record(); torch.cuda.synchronize()
        res[tag] = 1e3 * e0.elapsed_time(e1) / iters
    print(f"temporal attention {name} (F={F_} P={P} H={H} T={T}): fwd {res['fwd']:7.1f} us  bwd {res['bwd']:7.1f} us", flush=True)
