"""Repeat the round-4 LDS-DMA kernels (stem forward / weight gradient, row-item 3x3 weight gradient) and the attention kernels with the LDS-patch stores many times,
with a second stream keeping the GPU busy, and compare every result with the first one: a missing wait / barrier in a double-buffered DMA pipeline shows up as an
occasional outlier, not in a single parity run.  usage: stress_new_kernels.py [rounds]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd import ops, _lib as L
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = "cuda"
torch.manual_seed(0)
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
# operands
x = torch.randn(64, 3, 224, 224, device=dev)
w = cl((torch.randn(64, 3, 7, 7, device=dev) * 147 ** -0.5).bfloat16())
xp = ops.stem_input(x, torch.bfloat16, 7, 2, own=True)
dy_s = cl(torch.randn(64, 64, 112, 112, device=dev).bfloat16())
xa = cl(torch.randn(64, 64, 56, 56, device=dev).bfloat16()); dya = cl(torch.randn(64, 64, 56, 56, device=dev).bfloat16())
qkv = torch.randn(64, 197, 1536, device=dev).bfloat16(); do = torch.randn(64, 197, 512, device=dev).bfloat16()
qkt = torch.randn(64, 197, 1536, device=dev).bfloat16()
busy_a = torch.randn(4096, 4096, device=dev).bfloat16(); busy_b = torch.randn(4096, 4096, device=dev).bfloat16()
side = torch.cuda.Stream()

def run():
    out = {}
    sums = torch.zeros(64, 32, 2, dtype=torch.float64, device=dev)
    dw = torch.zeros(64, 147, device=dev)
    y = ops.StemConvFn.apply(xp, w.requires_grad_(True), dw, sums, (224, 224))
    y.backward(dy_s)
    ops.side_stream_join(torch.device(dev))
    out["stem y"], out["stem sums"], out["stem dw"] = y.detach().float(), sums.float(), dw
    out["rows dw"] = ops.conv3x3_wgrad(dya, xa).clone()
    o, lse = ops.attn_spatial_fwd(qkv, 8)
    out["attn o"], out["attn dqkv"] = o.float(), ops.attn_spatial_bwd(qkv, o, do, lse, 8).float()
    ot, lt = ops.attn_temporal_fwd(qkt, 8, 16)
    out["tm o"], out["tm dqkv"] = ot.float(), ops.attn_temporal_bwd(qkt, ot, do, lt, 8, 16).float()
    torch.cuda.synchronize()
    return out

ref = run()
worst = {k: 0.0 for k in ref}
for it in range(rounds):
    with torch.cuda.stream(side):
        for _ in range(6):
            busy_a @ busy_b
    got = run()
    for k in ref:
        d = float((got[k] - ref[k]).abs().max() / (ref[k].abs().max() + 1e-30))
        worst[k] = max(worst[k], d)
        assert torch.isfinite(got[k]).all(), (it, k)
torch.cuda.synchronize()
print(f"{rounds} rounds under a busy second stream; worst relative deviation from the first run per result:")
for k, v in worst.items():
    print(f"  {k:12s} {v:.3e}")
bad = [k for k, v in worst.items() if v > 2e-3]
print("OK" if not bad else f"OUTLIERS: {bad}")
sys.exit(1 if bad else 0)
