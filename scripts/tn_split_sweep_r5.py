"""split-M target sweep of the LDS-DMA weight-gradient kernel (MAED_OPT_TN_TARGET_WGS; 0 = the heuristic tuned on the register-transposing kernel in round 3)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd import ops, _lib as L
M = 128 * 197
shapes = [("qkv", M, 1536, 512), ("fc1", M, 2048, 512), ("fc2", M, 512, 2048), ("proj", M, 512, 512), ("c1 56 64>256", 401408, 256, 64), ("c2 28 128>512", 100352, 512, 128),
          ("c3 14 256>1024", 25088, 1024, 256), ("c3 14 1024>256", 25088, 256, 1024), ("c3 s2 512>1024", 25088, 1024, 512)]
targets = [0, 128, 192, 256, 320, 384, 512, 640, 768, 1024]
L.set_option(L.OPT_TN_DMA, 1)
print("shape".ljust(18) + "".join(f"{t:>8d}" for t in targets))
for name, m, n, k in shapes:
    Y = [torch.randn(m, n, device="cuda").bfloat16() for _ in range(2)]
    X = [torch.randn(m, k, device="cuda").bfloat16() for _ in range(2)]
    row = []
    for t in targets:
        L.set_option(L.OPT_TN_TARGET_WGS, t)
        dW = torch.zeros(n, k, device="cuda"); db = torch.zeros(n, device="cuda")
        best = 1e9
        for r in range(3):
            ops.gemm_tn_wgrad(Y[0], X[0], dW=dW, dbias=db); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(20):
                ops.gemm_tn_wgrad(Y[i & 1], X[i & 1], dW=dW, dbias=db)
            e1.record(); torch.cuda.synchronize()
            best = min(best, 1e3 * e0.elapsed_time(e1) / 20)
        row.append(best)
    print(name.ljust(18) + "".join(f"{v:8.1f}" for v in row), flush=True)
L.set_option(L.OPT_TN_TARGET_WGS, 0)
