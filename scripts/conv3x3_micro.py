"""3x3 convolutions of the hybrid R50 at cfg3 (128 frames, bf16 channels_last): MIOpen through ATen (including the output zero-fill
its solvers enqueue) vs maed_conv3x3_fwd (implicit GEMM, gathered A rows).  Forward for all 16, input gradient for the 13 stride-1
ones (the same kernel on dY with the flipped, transposed weights).  The own kernel has not run on hardware yet (round-2 first call)."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd import ops
from maed_amd.resnetv2 import _same_pad
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
Fr = 128
#          H_in  C   stride count
shapes = [(56, 64, 1, 3), (56, 128, 2, 1), (28, 128, 1, 3), (28, 256, 2, 1), (14, 256, 1, 8)]


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


tot = dict(fwd_miopen=0.0, fwd_own=0.0, dgrad_miopen=0.0, dgrad_own=0.0, wgrad_miopen=0.0, wgrad_own=0.0)
for H, C, s, cnt in shapes:
    x = torch.randn(Fr, C, H, H, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, C, 3, 3, device="cuda") * (9 * C) ** -0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    w_taps = w.permute(0, 2, 3, 1)                      # (O,3,3,I) contiguous view of the channels_last weight
    w_flip = w.flip(2, 3).permute(1, 2, 3, 0).contiguous()
    ref = F.conv2d(_same_pad(x, 3, s), w, None, s)
    dy = torch.randn_like(ref)
    pad = 1 if s == 1 else 0
    xin = x if s == 1 else _same_pad(x, 3, s)
    fm = timeit(lambda: F.conv2d(xin, w, None, s, pad))
    fo = timeit(lambda: ops.conv3x3(x, w_taps, s))
    y = ops.conv3x3(x, w_taps, s)
    err = ((y.float() - ref.float()).abs().max() / ref.float().abs().max()).item()
    line = f"H={H:3d} C={C:4d} stride {s} x{cnt}: fwd miopen {fm:7.1f} own {fo:7.1f} us  rel err {err:.1e}"
    tot["fwd_miopen"] += cnt * fm; tot["fwd_own"] += cnt * fo
    if s == 1:
        dm = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0])
        do = timeit(lambda: ops.conv3x3(dy, w_flip, 1))
        gref = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        gerr = ((ops.conv3x3(dy, w_flip, 1).float() - gref.float()).abs().max() / gref.float().abs().max()).item()
        line += f" | dgrad miopen {dm:7.1f} own {do:7.1f} us  rel err {gerr:.1e}"
        tot["dgrad_miopen"] += cnt * dm; tot["dgrad_own"] += cnt * do
        wm = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1])
        wo = timeit(lambda: ops.conv3x3_wgrad(dy, x))
        wref = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1].float()
        werr = ((ops.conv3x3_wgrad(dy, x).permute(0, 3, 1, 2) - wref).abs().max() / wref.abs().max()).item()
        line += f" | wgrad miopen {wm:7.1f} own {wo:7.1f} us  rel err {werr:.1e}"
        tot["wgrad_miopen"] += cnt * wm; tot["wgrad_own"] += cnt * wo
    print(line, flush=True)
print("backbone totals (ms): " + "  ".join(f"{k} {v / 1e3:.2f}" for k, v in tot.items()))
