"""Whole-head vs K/V-tiled attention kernels on the GPU (round-2 kick-off measurement; nothing here has run on hardware yet --
the tiled kernels of csrc/attn_long.hip were developed on the host simulator after the round-1 GPU budget was spent).

  * spatial shapes (cfg3 P=197, cfg5 P=257): whole-head MFMA kernels (impl 2) vs tiled kernels (impl 5), forward and backward,
    plus a parity check of the two against each other;
  * coupling shape (N=8 clips x 16 x 197 = 3152 tokens, H=8): tiled kernels only (the whole-head kernels cannot hold it).
Usage: python scripts/attn_long_micro.py [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


def run(tag, F_, L_, H, impls):
    C = 64 * H
    torch.manual_seed(0)
    qkv = torch.randn(F_, L_, 3 * C, device="cuda").bfloat16()
    do = torch.randn(F_, L_, C, device="cuda").bfloat16()
    fl_f, fl_b = 4.0 * L_ * L_ * C * F_, 10.0 * L_ * L_ * C * F_
    outs = {}
    for impl in impls:
        o, lse = ops.attn_spatial_fwd(qkv, H, impl)
        g = ops.attn_spatial_bwd(qkv, o, do, lse, H, impl=impl)
        outs[impl] = (o.float(), lse, g.float())
        uf = timed(lambda: ops.attn_spatial_fwd(qkv, H, impl))
        ub = timed(lambda: ops.attn_spatial_bwd(qkv, o, do, lse, H, impl=impl))
        print(f"{tag:28s} impl={impl}: fwd {uf:9.1f} us ({fl_f / uf / 1e6:7.1f} TF/s)   bwd {ub:9.1f} us ({fl_b / ub / 1e6:7.1f} TF/s)", flush=True)
    if len(impls) == 2:
        a, b = outs[impls[0]], outs[impls[1]]
        print(f"{'':28s} max |diff| o {float((a[0] - b[0]).abs().max()):.3e}  lse {float((a[1] - b[1]).abs().max()):.3e}  "
              f"dqkv {float((a[2] - b[2]).abs().max()):.3e} (ref max {float(a[2].abs().max()):.3e})")


run("spatial cfg3 F128 P197 H8", 128, 197, 8, [2, 5])
run("spatial cfg5 F128 P257 H12", 128, 257, 12, [2, 5])
run("coupling cfg3 N8 L3152 H8", 8, 16 * 197, 8, [0])
