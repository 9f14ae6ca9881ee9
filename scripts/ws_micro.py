"""Batched weight standardisation of the hybrid R50's 53 convolutions (maed_weight_std_fwd / _bwd through ops.WeightStdFn), bf16 mode; usage: ws_micro.py [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MAED_SYNTHETIC_SMPL_OK", "1")
from maed_amd import ops
from maed_amd.resnetv2 import ResNetV2
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
net = ResNetV2(compute_dtype=torch.bfloat16).cuda()
ws = [w for w in net.conv_weights()]
def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters
def fwd():
    with torch.no_grad():
        return ops.WeightStdFn.apply(net, torch.bfloat16, 1e-5, *ws)
print(f"weight standardisation forward (53 convolutions, {sum(w.numel() for w in ws) / 1e6:.1f} M weights, incl. transposed images): {timed(fwd):7.1f} us")
outs = ops.WeightStdFn.apply(net, torch.bfloat16, 1e-5, *ws)
g = [torch.randn_like(o) for o in outs]
def fb():
    o = ops.WeightStdFn.apply(net, torch.bfloat16, 1e-5, *ws)
    torch.autograd.backward(o, g)
print(f"forward + backward (gradients through autograd for every convolution): {timed(fb):7.1f} us")
