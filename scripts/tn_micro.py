"""bf16 weight-gradient GEMM dW += Y^T X at the cfg3 shapes (STE linears, backbone 1x1 convolutions) in isolation: the register-transposing kernel (gemm_tn.hip,
MAED_OPT_TN_DMA = 0) against the LDS-DMA + transposing-read kernel (gemm_tn2.hip, = 1), interleaved rounds in one process, + a correctness check of both against
an fp32 product of the same bf16 operands.   usage: tn_micro.py [iters] [rounds]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd import ops, _lib as L
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.manual_seed(0)
M = 128 * 197
shapes = [("qkv", M, 1536, 512), ("fc1", M, 2048, 512), ("fc2", M, 512, 2048), ("proj", M, 512, 512), ("ts", 128, 1024, 1024),
          ("c1 56 64>256", 401408, 256, 64), ("c1 56 256>64", 401408, 64, 256), ("c2 28 128>512", 100352, 512, 128), ("c2 28 512>128", 100352, 128, 512),
          ("c3 14 256>1024", 25088, 1024, 256), ("c3 14 1024>256", 25088, 256, 1024), ("c2 28 256>512 s2", 100352, 512, 256), ("c3 14 512>1024 s2", 25088, 1024, 512)]
MODES = (0, 2, 3)      # MAED_OPT_TN_DMA: 0 = register-transposing kernel, 2 / 3 = LDS-DMA kernel with 128 x 128 / 256 x 256 tiles forced
tot = {d: 0.0 for d in MODES}
for name, m, n, k in shapes:
    Y = [torch.randn(m, n, device="cuda").bfloat16() for _ in range(2)]
    X = [torch.randn(m, k, device="cuda").bfloat16() for _ in range(2)]
    res = {}
    best = {d: 1e9 for d in MODES}
    for r in range(rounds):
        for dma in MODES:
            L.set_option(L.OPT_TN_DMA, dma)
            dW = torch.zeros(n, k, device="cuda"); db = torch.zeros(n, device="cuda")
            ops.gemm_tn_wgrad(Y[0], X[0], dW=dW, dbias=db)
            if r == 0:
                res[dma] = (dW.clone(), db.clone())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                ops.gemm_tn_wgrad(Y[i & 1], X[i & 1], dW=dW, dbias=db)
            e1.record(); torch.cuda.synchronize()
            best[dma] = min(best[dma], 1e3 * e0.elapsed_time(e1) / iters)
    ref = Y[0].float().t() @ X[0].float()
    refb = Y[0].float().sum(0)
    err = {d: max(float((res[d][0] - ref).abs().max() / ref.abs().max()), float((res[d][1] - refb).abs().max() / refb.abs().max())) for d in MODES}
    for d in MODES:
        tot[d] += best[d]
    print(f"tn {name:18s} M={m} N={n} K={k}: perm {best[0]:7.1f} us ({2.0 * m * n * k / best[0] / 1e6:6.0f} TF)   dma 128^2 {best[2]:7.1f} us ({2.0 * m * n * k / best[2] / 1e6:6.0f} TF)   "
          f"dma 256^2 {best[3]:7.1f} us ({2.0 * m * n * k / best[3] / 1e6:6.0f} TF)   rel err {err[0]:.1e} {err[2]:.1e} {err[3]:.1e}", flush=True)
print("sum over shapes: " + ", ".join(f"mode {d}: {tot[d]:.1f} us" for d in MODES))
