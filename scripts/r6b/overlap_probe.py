"""Round 6 (second session): do the STE's weight-gradient GEMMs (L2 -> LDS / MFMA side) run beside the backbone backward's HBM-bound kernels for free?
Two streams, one kernel family each, alone and together: if together ~ max(alone) the two are complementary and the STE's weight gradients could be deferred
to run under the backbone's backward; if together ~ sum, nothing is to be had.   usage: overlap_probe.py [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from maed_amd import ops, _lib as L
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
lib = L.lib()
p = lambda t: None if t is None else t.data_ptr()
torch.manual_seed(0)
dev = "cuda"
M = 128 * 197
# stream A: the five weight gradients of one STE block (cfg3)
tn = [(M, 512, 2048), (M, 2048, 512), (M, 512, 512), (M, 1536, 512)]
tnY = [torch.randn(m, n, device=dev).bfloat16() for m, n, k in tn]
tnX = [torch.randn(m, k, device=dev).bfloat16() for m, n, k in tn]
tnW = [torch.zeros(n, k, device=dev) for m, n, k in tn]
tnB = [torch.zeros(n, device=dev) for m, n, k in tn]
def run_tn():
    for i in range(len(tn)):
        ops.gemm_tn_wgrad(tnY[i], tnX[i], dW=tnW[i], dbias=tnB[i])
# stream B candidates: GroupNorm backward (stage 2 residual shape), backbone 1x1 dgrad GEMM (stage 1: M = 401408, 256 -> 64), STE dgrad GEMM (dfc1)
N, HW, C = 128, 28 * 28, 512
gx = torch.randn(N, HW, C, device=dev).bfloat16(); gdy = torch.randn(N, HW, C, device=dev).bfloat16(); gdx = torch.empty_like(gx)
gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
mask = torch.randint(0, 256, (N * HW * C // 8,), dtype=torch.uint8, device=dev)
sums = torch.zeros(N, 32, 2, dtype=torch.float64, device=dev)
v = gx.float().view(N, HW, 32, C // 32).double()
sums[:, :, 0] = v.sum((1, 3)); sums[:, :, 1] = (v * v).sum((1, 3))
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
ab = torch.zeros(N * C * 2 + N * ops.GN_SYNC_WORDS, device=dev)
def run_gn():
    for _ in range(4):
        ab.zero_()
        ops.check(lib.maed_groupnorm_bwd(p(gx), p(mask), p(gdy), p(sums), p(gamma), p(beta), p(gdx), None, p(dg), p(db), p(ab), N, HW, C, 1e-5, 1,
                                         ops.dt_code(torch.bfloat16), 1, p(ab) + 4 * N * C * 2, None, ops._stream()), "gn bwd")
A1 = torch.randn(401408, 256, device=dev).bfloat16(); W1 = torch.randn(64, 256, device=dev).bfloat16()
def run_c1():
    for _ in range(4):
        ops.gemm_nt(A1, W1, L.EPI_STORE)
A2 = torch.randn(M, 2048, device=dev).bfloat16(); W2 = torch.randn(512, 2048, device=dev).bfloat16()
def run_dfc1():
    for _ in range(4):
        ops.gemm_nt(A2, W2, L.EPI_STORE)
side = torch.cuda.Stream()
def timed(fa, fb):
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if fa is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(iters):
                    fa()
        if fb is not None:
            for _ in range(iters):
                fb()
        if fa is not None:
            torch.cuda.current_stream().wait_stream(side)
        e1.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / iters)
    return best
for f in (run_tn, run_gn, run_c1, run_dfc1):
    f()
torch.cuda.synchronize()
ta = timed(run_tn, None)
print(f"STE weight gradients of one block (fc2, fc1, proj, qkv) alone: {ta:7.1f} us")
for name, fb in (("4 x GroupNorm backward, stage-2 residual layer", run_gn), ("4 x stage-1 1x1 convolution 256 -> 64 (NT GEMM, HBM-bound)", run_c1), ("4 x STE input-gradient GEMM d(fc1)", run_dfc1)):
    tb = timed(None, fb)
    tab = timed(run_tn, fb)
    print(f"{name}: alone {tb:7.1f} us   together {tab:7.1f} us   sum {ta + tb:7.1f}   max {max(ta, tb):7.1f}   hidden {100 * (ta + tb - tab) / min(ta, tb):5.1f} % of the shorter one")
