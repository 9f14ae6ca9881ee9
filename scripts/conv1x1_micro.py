"""1x1 convolutions of the hybrid R50 at cfg3 (128 frames, bf16 channels_last): MIOpen through ATen (including the
output zero-fill / workspace casts its solvers enqueue) vs maed_gemm_nt on the (N*H*W, C) matrix view.
Forward: Y = X W^T (W (O,I)); input gradient: dX = dY Wt^T (Wt (I,O))."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd import ops, _lib as L
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
Fr = 128
shapes = [(56, 64, 64, 1), (56, 64, 256, 4), (56, 256, 64, 2), (56, 256, 128, 1), (28, 128, 512, 4), (28, 512, 128, 3), (28, 512, 256, 1),
          (14, 256, 1024, 9), (14, 1024, 256, 8)]


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


tot = dict(fm=0.0, fg=0.0, dm=0.0, dg=0.0)
for H, I, O, cnt in shapes:
    x = torch.randn(Fr, I, H, H, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    dy = torch.randn(Fr, O, H, H, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(O, I, 1, 1, device="cuda") * I ** -0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    X, Y = x.permute(0, 2, 3, 1).reshape(-1, I), dy.permute(0, 2, 3, 1).reshape(-1, O)
    W2, Wt = w.reshape(O, I).contiguous(), w.reshape(O, I).t().contiguous()
    yo, dxo = torch.empty(X.shape[0], O, device="cuda", dtype=torch.bfloat16), torch.empty(X.shape[0], I, device="cuda", dtype=torch.bfloat16)
    fm = timeit(lambda: F.conv2d(x, w))
    fg = timeit(lambda: ops.gemm_nt(X, W2, L.EPI_STORE, out=yo))
    dm = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0])
    dg = timeit(lambda: ops.gemm_nt(Y, Wt, L.EPI_STORE, out=dxo))
    ref = F.conv2d(x, w).permute(0, 2, 3, 1).reshape(-1, O).float()
    err = ((yo.float() - ref).abs().max() / ref.abs().max()).item()
    for k, v in (("fm", fm), ("fg", fg), ("dm", dm), ("dg", dg)):
        tot[k] += cnt * v
    print(f"H={H:3d} I={I:4d} O={O:4d} x{cnt}: fwd miopen {fm:7.1f} gemm {fg:7.1f} | dgrad miopen {dm:7.1f} gemm {dg:7.1f} us   fwd rel err {err:.1e}", flush=True)
print("backbone totals (stride-1 1x1 convs, ms): " + "  ".join(f"{k} {v / 1e3:.2f}" for k, v in tot.items()))
