"""1x1-convolution weight gradients of the hybrid R50 at cfg3 (128 frames): MIOpen (through ATen, incl. its workspace
zero/cast passes) vs maed_gemm_tn_wgrad on the channels_last activations viewed as (N*H*W, C) matrices."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd import ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
Fr = 128
# (H, I, O, count in the backbone)
M_STE = 128 * 197
ste = [("qkv", 1536, 512), ("fc1", 2048, 512), ("fc2", 512, 2048), ("proj", 512, 512)]
if os.environ.get("WGRAD_STE", "0") == "1":
    tot = 0.0
    for name, N, K in ste:
        Y = torch.randn(M_STE, N, device="cuda").bfloat16(); X = torch.randn(M_STE, K, device="cuda").bfloat16()
        dW = torch.zeros(N, K, device="cuda")
        for _ in range(3):
            ops.gemm_tn_wgrad(Y, X, dW=dW)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.gemm_tn_wgrad(Y, X, dW=dW)
        e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / iters
        tot += us
        print(f"STE wgrad {name:5s} N={N} K={K}: {us:7.1f} us  {2.0 * M_STE * N * K / us / 1e6:6.1f} TF", flush=True)
    print(f"STE wgrad per block: {tot:.1f} us")
shapes = [(56, 64, 64, 1), (56, 64, 256, 4), (56, 256, 64, 2), (56, 256, 128, 1), (28, 128, 512, 4), (28, 512, 128, 3), (28, 512, 256, 1),
          (14, 256, 1024, 9), (14, 1024, 256, 8)]
tot_m = tot_g = 0.0
for H, I, O, cnt in shapes:
    x = torch.randn(Fr, I, H, H, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    dy = torch.randn(Fr, O, H, H, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(O, I, 1, 1, device="cuda").bfloat16()
    def miopen():
        return torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    X = x.permute(0, 2, 3, 1).reshape(-1, I)
    Y = dy.permute(0, 2, 3, 1).reshape(-1, O)
    dW = torch.zeros(O, I, device="cuda")
    def mine():
        return ops.gemm_tn_wgrad(Y, X, dW=dW)
    res = {}
    for name, fn in (("miopen", miopen), ("gemm_tn", mine)):
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            out = fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = 1e3 * e0.elapsed_time(e1) / iters
    dW.zero_(); mine(); torch.cuda.synchronize()
    ref = miopen().float().reshape(O, I)
    err = ((dW - ref).abs().max() / ref.abs().max()).item()
    tot_m += cnt * res["miopen"]; tot_g += cnt * res["gemm_tn"]
    print(f"H={H:3d} I={I:4d} O={O:4d} x{cnt}: miopen {res['miopen']:7.1f} us   gemm_tn {res['gemm_tn']:7.1f} us   rel err {err:.2e}", flush=True)
print(f"backbone total (stride-1 1x1 convs): miopen {tot_m / 1e3:.2f} ms   gemm_tn {tot_g / 1e3:.2f} ms")
