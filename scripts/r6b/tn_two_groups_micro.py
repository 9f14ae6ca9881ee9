"""Round 6 (second session): the weight-gradient kernel with TWO wave groups per workgroup (gemm_tn2.hip, G = 2, MAED_OPT_TN_DMA = 5) against the default (= 1) at the
cfg3 shapes, interleaved rounds in one process, with a split sweep of the new form (MAED_OPT_TN_TARGET_WGS).   usage: tn_two_groups_micro.py [iters] [rounds]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from maed_amd import ops, _lib as L
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.manual_seed(0)
M = 128 * 197
shapes = [("qkv", M, 1536, 512), ("fc1", M, 2048, 512), ("fc2", M, 512, 2048), ("proj", M, 512, 512), ("ts", 128, 1024, 1024),
          ("c1 56 64>256", 401408, 256, 64), ("c1 56 256>64", 401408, 64, 256), ("c2 28 128>512", 100352, 512, 128), ("c2 28 512>128", 100352, 128, 512),
          ("c3 14 256>1024", 25088, 1024, 256), ("c3 14 1024>256", 25088, 256, 1024), ("c2 28 256>512 s2", 100352, 512, 256), ("c3 14 512>1024 s2", 25088, 1024, 512),
          ("cfg5 fc2", 32896, 768, 3072), ("cfg5 qkv", 32896, 2304, 768)]
MODES = [(1, 0), (5, 0), (5, 128), (5, 192), (5, 320), (5, 384), (5, 512)]     # (MAED_OPT_TN_DMA, MAED_OPT_TN_TARGET_WGS)
tot = {d: 0.0 for d in MODES}
L.set_option(L.OPT_TN_SK, 0)
for name, m, n, k in shapes:
    Y = [torch.randn(m, n, device="cuda").bfloat16() for _ in range(2)]
    X = [torch.randn(m, k, device="cuda").bfloat16() for _ in range(2)]
    res = {}
    best = {d: 1e9 for d in MODES}
    for r in range(rounds):
        for d in MODES:
            L.set_option(L.OPT_TN_DMA, d[0]); L.set_option(L.OPT_TN_TARGET_WGS, d[1])
            dW = torch.zeros(n, k, device="cuda"); db = torch.zeros(n, device="cuda")
            ops.gemm_tn_wgrad(Y[0], X[0], dW=dW, dbias=db)
            if r == 0:
                res[d] = (dW.clone(), db.clone())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                ops.gemm_tn_wgrad(Y[i & 1], X[i & 1], dW=dW, dbias=db)
            e1.record(); torch.cuda.synchronize()
            best[d] = min(best[d], 1e3 * e0.elapsed_time(e1) / iters)
    ref = Y[0].float().t() @ X[0].float()
    refb = Y[0].float().sum(0)
    err = {d: max(float((res[d][0] - ref).abs().max() / ref.abs().max()), float((res[d][1] - refb).abs().max() / refb.abs().max())) for d in MODES}
    for d in MODES:
        tot[d] += best[d]
    print(f"tn {name:18s} M={m} N={n} K={k}: " + "  ".join(f"{d[0]}/{d[1]}: {best[d]:6.1f} us" for d in MODES) + f"   rel err max {max(err.values()):.1e}", flush=True)
print("sum over shapes: " + ", ".join(f"{d[0]}/{d[1]}: {tot[d]:.1f} us" for d in MODES))
