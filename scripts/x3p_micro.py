"""The accurate mode's forward products at the cfg3 STE shapes in isolation: the register-staged split kernel on fp32 operands (gemm_x3.hip, maed_gemm_nt with
MAED_F32X3) against the LDS-DMA kernel on operands stored as (hi, lo) bf16 planes (gemm_x3p.hip, maed_gemm_nt_planes) with a 2- and a 4-stage copy ring --
interleaved rounds in one process, + bit-for-bit agreement of the three.   usage: x3p_micro.py [iters] [rounds]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd import ops, _lib as L
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.manual_seed(0)
M = 128 * 197
shapes = [("qkv", M, 1536, 512, L.EPI_STORE), ("fc1+gelu", M, 2048, 512, L.EPI_GELU), ("fc2+res", M, 512, 2048, L.EPI_RESID_F32), ("proj+res", M, 512, 512, L.EPI_RESID_F32),
          ("c2 28 512>128", 100352, 128, 512, L.EPI_STORE), ("c3 14 256>1024", 25088, 1024, 256, L.EPI_STORE), ("c3 14 1024>256", 25088, 256, 1024, L.EPI_STORE)]
VAR = {"p2": 2, "p4": 4, "p5": 5, "p6": 6, "p7": 7}      # 128^2 x 2 stages, 128^2 x 4 stages, 256x128 x 3 stages, 256^2 x 2 stages
tot = {k: 0.0 for k in ("x3", "p2", "p4", "p5", "p6", "p7")}
for name, m, n, k, epi in shapes:
    A = [torch.randn(m, k, device="cuda") for _ in range(2)]
    B = torch.randn(n, k, device="cuda") * k ** -0.5
    bias = torch.randn(n, device="cuda")
    res = torch.randn(m, n, device="cuda") if epi == L.EPI_RESID_F32 else None
    Ap = [ops.split_planes(a) for a in A]
    Bp = ops.split_planes(B)
    gelu = epi == L.EPI_GELU

    def run(which, i):
        if which == "x3":
            o = ops.gemm_nt(A[i], B, epi, bias=bias, aux=res, prec="bf16x3")
            return o[0] if gelu else o
        # what the twin forward asks of fc1: activation as planes only + the pre-activation as bf16; of the others: the fp32 result
        o, pl, pre = ops.gemm_nt_planes(Ap[i], Bp, epi, bias=bias, aux=res, want_f32=not gelu, want_planes=gelu, want_pre=gelu, variant=VAR[which])
        return pl if gelu else o

    out = {w: run(w, 0) for w in tot}
    torch.cuda.synchronize()
    if gelu:
        hi = out["x3"].bfloat16(); lo = (out["x3"] - hi.float()).bfloat16()
        same = all(torch.equal(out[w][0], hi) and torch.equal(out[w][1], lo) for w in VAR)
    else:
        same = all(torch.equal(out["x3"], out[w]) for w in VAR)
    del out
    best = {w: 1e9 for w in tot}
    for r in range(rounds):
        for w in tot:
            run(w, 0); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                run(w, i & 1)
            e1.record(); torch.cuda.synchronize()
            best[w] = min(best[w], 1e3 * e0.elapsed_time(e1) / iters)
    for w in tot:
        tot[w] += best[w]
    fl = 3 * 2.0 * m * n * k
    print(f"nt {name:16s} M={m} N={n} K={k}: fp32 operands {best['x3']:7.1f} us ({fl / best['x3'] / 1e6:5.0f} TF of bf16 MFMA) | planes 128^2 x2 {best['p2']:7.1f} ({fl / best['p2'] / 1e6:5.0f} TF)  "
          f"128^2 x4 {best['p4']:7.1f}  256x128 x3 {best['p5']:7.1f} ({fl / best['p5'] / 1e6:5.0f} TF)  256^2 x2 {best['p6']:7.1f} ({fl / best['p6'] / 1e6:5.0f} TF)  128^2 k64 {best['p7']:7.1f} ({fl / best['p7'] / 1e6:5.0f} TF) | bit-identical: {same}", flush=True)
print("sum over shapes: " + ", ".join(f"{w}: {tot[w]:.1f} us" for w in tot) + "   (timings include the output allocation of each call: the same for all three)")
