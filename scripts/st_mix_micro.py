"""st_colmean / st_mix_bwd_reduce at the cfg3 token geometry (128 frames x 197 tokens x 512 channels), event timing.  python scripts/st_mix_micro.py [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd import _lib as L, ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda")
for dt in (torch.bfloat16, torch.float32):
    F_, P, C = 128, 197, 512
    xs, xt, dm = (torch.randn(F_, P, C, device=dev).to(dt) for _ in range(3))
    logits = torch.randn(F_, 2 * C, device=dev)
    means = torch.empty(F_, 2 * C, device=dev, dtype=dt); ws = torch.empty(F_, 2 * C, device=dev)
    dlog = torch.empty(F_, 2 * C, device=dev, dtype=dt)
    lib, p, st = L.lib(), ops._p, torch.cuda.current_stream().cuda_stream
    def t(fn):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / iters
    a = t(lambda: ops.check(lib.maed_st_colmean(p(xs), p(xt), p(means), p(ws), F_, P, C, ops.dt_code(dt), st), "colmean"))
    b = t(lambda: ops.check(lib.maed_st_mix_bwd_reduce(p(dm), p(xs), p(xt), p(logits), p(dlog), p(ws), F_, P, C, ops.dt_code(dt), st), "reduce"))
    print(f"{dt}: st_colmean {a:6.1f} us   st_mix_bwd_reduce {b:6.1f} us   (3 launches each: memset, reduction, finish)   checks {means.float().abs().sum().item():.4e} {dlog.float().abs().sum().item():.4e}", flush=True)
