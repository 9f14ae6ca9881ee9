"""Repeat the round-6 persistent K-stream GEMMs (csrc/gemm_sk.hip: copy ring running through item boundaries, epilogue staging inside the ring, slab hand-over between
workgroups with flags; csrc/gemm_tn_sk.hip: slabs + reduce launch) many times while a SECOND stream keeps the chip busy -- alternately with plain GEMMs and with another
persistent kernel of the same family (its own slab set; two such kernels share the CUs, so a finisher's predecessor may not be resident yet: the case the hand-over
protocol is designed for) -- into NaN-filled outputs, every result compared bit for bit with the first: a copy landing late, a ring slot re-targeted early, a stale or
half-written slab shows up as an occasional differing launch, not in a single parity run.   usage: stress_r6_kernels.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd import ops, _lib as L  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = "cuda"
lib = L.lib()
torch.manual_seed(0)
M = 128 * 197
nt_shapes = [(M, 1536, 512, L.EPI_STORE), (M, 512, 2048, L.EPI_RESID_F32), (M, 2048, 512, L.EPI_GELU), (32896, 768, 3072, L.EPI_STORE), (3000, 520, 256, L.EPI_STORE)]
tn_shapes = [(M, 1536, 512), (32896, 768, 3072), (25088, 1024, 256), (1280, 264, 392)]
side = torch.cuda.Stream()
busy_a, busy_b = torch.randn(4096, 4096, device=dev).bfloat16(), torch.randn(4096, 4096, device=dev).bfloat16()
sA, sB = torch.randn(M, 2048, device=dev).bfloat16(), (torch.randn(512, 2048, device=dev) * 2048 ** -0.5).bfloat16()
sY, sX = torch.randn(M, 512, device=dev).bfloat16(), torch.randn(M, 2048, device=dev).bfloat16()
cases = []
for m, n, k, epi in nt_shapes:
    A, B = torch.randn(m, k, device=dev).bfloat16(), (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
    bias = torch.randn(n, device=dev)
    aux = torch.randn(m, n, device=dev) if epi == L.EPI_RESID_F32 else None
    cases.append(("nt", (A, B, bias, aux, epi)))
for m, n, k in tn_shapes:
    cases.append(("tn", (torch.randn(m, n, device=dev).bfloat16(), torch.randn(m, k, device=dev).bfloat16())))


def run(kind, c, mode):
    if kind == "nt":
        A, B, bias, aux, epi = c
        odt = torch.float32 if epi == L.EPI_RESID_F32 else torch.bfloat16
        out = torch.full((A.shape[0], B.shape[0]), float("nan"), device=dev, dtype=odt)
        out2 = torch.full_like(out, float("nan")) if epi == L.EPI_GELU else None
        lib.maed_set_option(L.OPT_SK, mode)
        ops.gemm_nt(A, B, epi, bias=bias, out=out, out2=out2, aux=aux, impl=L.IMPL_MFMA_SK)
        return (out,) if out2 is None else (out, out2)
    Y, X = c
    dW, db = torch.zeros(Y.shape[1], X.shape[1], device=dev), torch.zeros(Y.shape[1], device=dev)
    lib.maed_set_option(L.OPT_SK_GRID, 255)       # (an explicit grid takes the persistent weight-gradient kernel whatever the heuristic says)
    ops.gemm_tn_wgrad(Y, X, dW=dW, dbias=db)
    lib.maed_set_option(L.OPT_SK_GRID, 0)
    return dW, db


first = {}
bad = 0
for it in range(rounds):
    with torch.cuda.stream(side):
        if it % 2 == 0:
            for _ in range(4):
                busy_a @ busy_b
        else:                                       # another persistent kernel pair on the side stream (its own slab set), sharing the CUs with the ones under test
            lib.maed_set_option(L.OPT_SK, 3)
            for _ in range(3):
                ops.gemm_nt(sA, sB, L.EPI_STORE, impl=L.IMPL_MFMA_SK)
            lib.maed_set_option(L.OPT_SK_GRID, 255)
            ops.gemm_tn_wgrad(sY, sX)
            lib.maed_set_option(L.OPT_SK_GRID, 0)
    for ci, (kind, c) in enumerate(cases):
        for mode in ((2, 3) if kind == "nt" else (1,)):
            got = run(kind, c, mode)
            key = (ci, mode)
            if key not in first:
                torch.cuda.synchronize()
                assert all(not torch.isnan(t.float()).any() for t in got), key
                first[key] = got
            else:
                bad += int(any(not torch.equal(a, b) for a, b in zip(got, first[key])))
    torch.cuda.synchronize()
lib.maed_set_option(L.OPT_SK, 1)
print(f"{rounds} rounds x {len(first)} (kernel, shape, mode) cases under a busy second stream (plain GEMMs / another persistent kernel alternately): {bad} differing launches")
print("device faults:", L.device_faults())
print("OK" if bad == 0 and L.device_faults() == 0 else "FAILED")
