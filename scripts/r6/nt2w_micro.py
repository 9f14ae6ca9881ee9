"""Round 6: the 256 x 128 / two-workgroups-per-CU NT GEMM (csrc/gemm2w.hip, impl = MAED_IMPL_MFMA_2W) against the dispatcher's choice among the per-tile kernels of
rounds 1-5 (128 x 128 at four workgroups per CU, 256 x 256 pipelined) at every NT shape of the step; bitwise against the 128 x 128 kernel (same k order), repeated
launches compared, five interleaved rounds with rotating operands.   usage: nt2w_micro.py [iters]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from maed_amd import ops, _lib as L  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M = 128 * 197
SHAPES = {"qkv": (M, 1536, 512, L.EPI_STORE), "fc1+gelu": (M, 2048, 512, L.EPI_GELU), "fc2+res": (M, 512, 2048, L.EPI_RESID_F32), "proj+res": (M, 512, 512, L.EPI_RESID_F32),
          "dfc2*gelu'": (M, 2048, 512, L.EPI_MUL_DGELU), "dqkv": (M, 512, 1536, L.EPI_STORE), "dfc1": (M, 512, 2048, L.EPI_STORE), "dproj": (M, 512, 512, L.EPI_STORE),
          "sq4k": (4096, 4096, 4096, L.EPI_STORE), "s1 64>256": (401408, 256, 64, L.EPI_STORE), "s1 256>64": (401408, 64, 256, L.EPI_STORE),
          "s2 128>512": (100352, 512, 128, L.EPI_STORE), "s2 512>128": (100352, 128, 512, L.EPI_STORE), "s3 256>1024": (25088, 1024, 256, L.EPI_STORE),
          "s3 1024>256": (25088, 256, 1024, L.EPI_STORE), "embed 1024>512": (25088, 512, 1024, L.EPI_STORE),
          "cfg5 qkv": (32896, 2304, 768, L.EPI_STORE), "cfg5 fc1+gelu": (32896, 3072, 768, L.EPI_GELU), "cfg5 fc2": (32896, 768, 3072, L.EPI_RESID_F32)}
lib = L.lib()
lib.maed_set_option(L.OPT_SK, 0)


def ev_time(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


print(f"# {'shape':16s} {'M x N x K':>20s} | dispatcher (rounds 1-5 kernels) | 256x128, 2 workgroups per CU | vendor bare   [us median of 5 rounds; TF]")
for name, (m, n, k, epi) in SHAPES.items():
    torch.manual_seed(0)
    A = [torch.randn(m, k, device="cuda").bfloat16() for _ in range(3)]
    B = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
    bias = torch.randn(n, device="cuda") if epi in (L.EPI_GELU, L.EPI_RESID_F32, L.EPI_STORE) else None
    aux = torch.randn(m, n, device="cuda") if epi == L.EPI_RESID_F32 else torch.randn(m, n, device="cuda").bfloat16() if epi == L.EPI_MUL_DGELU else None
    odt = torch.float32 if epi == L.EPI_RESID_F32 else torch.bfloat16
    out = torch.empty(m, n, device="cuda", dtype=odt)
    out2 = torch.empty_like(out) if epi == L.EPI_GELU else None
    vout = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    Bt = B.t()

    def run(i, impl, o=out, o2=out2):
        return ops.gemm_nt(A[i % 3], B, epi, bias=bias, out=o, out2=o2, aux=aux, impl=impl)

    want = torch.full_like(out, float("nan")); w2 = torch.full_like(out, float("nan")) if out2 is not None else None
    run(0, 3, want, w2)
    bad = 0
    for _ in range(10):
        o = torch.full_like(out, float("nan")); o2 = torch.full_like(out, float("nan")) if out2 is not None else None
        run(0, L.IMPL_MFMA_2W, o, o2)
        bad += int(not torch.equal(o, want) or (o2 is not None and not torch.equal(o2, w2)))
    ts = {"0": [], "w0": [], "w1": [], "w2": [], "w3": [], "v": []}
    for rnd_ in range(5):
        for key, impl in (("0", L.IMPL_AUTO), ("w0", L.IMPL_MFMA_2W), ("w1", L.IMPL_MFMA_2W), ("w2", L.IMPL_MFMA_2W), ("w3", L.IMPL_MFMA_2W)):
            lib.maed_set_option(L.OPT_SK_GRID, int(key[1]) if key != "0" else 0)       # (variant knob of the kernel while it is tuned)
            for i in range(2):
                run(i, impl)
            ts[key].append(ev_time(lambda i: run(i, impl), iters))
        lib.maed_set_option(L.OPT_SK_GRID, 0)
        for i in range(2):
            torch.mm(A[i % 3], Bt, out=vout)
        ts["v"].append(ev_time(lambda i: torch.mm(A[i % 3], Bt, out=vout), iters))
    fl = 2.0 * m * n * k / 1e6
    med = {k_: statistics.median(v) for k_, v in ts.items()}
    print(f"{name:16s} {m:7d}x{n:5d}x{k:5d} | {med['0']:7.1f} {fl / med['0']:7.1f} | 3st inter {med['w0']:6.1f}  3st front {med['w1']:6.1f}  2st inter {med['w2']:6.1f}  2st front {med['w3']:6.1f} | {med['v']:7.1f} {fl / med['v']:7.1f} |  "
          f"{'bit-equal to the 128x128 kernel in 10/10 launches' if bad == 0 else f'DIFFERS in {bad}/10 launches'}", flush=True)
    del A, B, out, vout, aux
lib.maed_set_option(L.OPT_SK, 1)
