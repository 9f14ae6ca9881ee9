"""STE spatial-attention forward at the cfg3 shape (F=128, P=197, H=8, bf16) in isolation: event timing and a
clean target for `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE` (HBM traffic per launch).  Inputs are ~N(0,1) random
(never zero-filled: DVFS / softmax work depend on the data)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd import ops
F_, P, H, C = 128, 197, 8, 512
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(0)
bufs = [torch.randn(F_, P, 3 * C, device="cuda").bfloat16() for _ in range(4)]   # rotate buffers: 4 x 77 MB > L2, < MALL
for q in bufs:
    ops.attn_spatial_fwd(q, H)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(iters):
    ops.attn_spatial_fwd(bufs[i % 4], H)
e1.record(); torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / iters
flops = 4.0 * P * P * C * F_
byts = 4.0 * F_ * P * C * 2 + 4.0 * F_ * H * P
print(f"attn_sp_fwd_mfma: {us:.2f} us/launch  {flops / us / 1e6:.1f} TFLOP/s ({flops / us / 1e6 / 2500:.3f} of MFMA peak)  "
      f"{byts / us / 1e3:.0f} GB/s algorithmic ({byts / us / 1e3 / 8000:.3f} of HBM peak)")
