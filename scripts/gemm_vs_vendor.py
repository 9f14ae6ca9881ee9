"""Yardstick: the library's bf16 GEMM kernels against the vendor library (torch.matmul -> hipBLASLt / rocBLAS) on the SAME box in the SAME
run, at every GEMM shape of the benchmarked step: the eight NT GEMMs of an STE block, the five TN weight-gradient shapes, the backbone's 1x1 convolutions (forward /
input-gradient NT shapes and their TN weight gradients).  Rotating operands (3 sets) so that no operand is L2 / MALL resident from the previous launch.

Two vendor columns.  (1) the PLAIN GEMM (no bias, no GELU, no residual, bf16 output; the TN products write a bf16 (N,K) matrix instead of accumulating in fp32)
against ours WITH its fused epilogue -- favours the vendor wherever ours does more.  (2) for the three epilogue shapes (bias+GELU with the pre-activation kept,
bias + fp32 residual, * GELU'), the vendor GEMM (bias through addmm, i.e. the vendor's own epilogue) followed by the ONE framework kernel that finishes what
ours fuses: like for like.  Yardstick only: nothing in maed_amd/ calls torch.matmul on a GPU tensor.

usage: gemm_vs_vendor.py [iters]      -> one line per shape; the last line names the shapes where the vendor path (2) is > 1.15x faster than ours"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd import ops, _lib as L  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
torch.manual_seed(0)
M = 128 * 197
NT = {"qkv": (M, 1536, 512, L.EPI_STORE), "fc1+gelu": (M, 2048, 512, L.EPI_GELU), "fc2+res": (M, 512, 2048, L.EPI_RESID_F32), "proj+res": (M, 512, 512, L.EPI_RESID_F32),
      "dfc2*gelu'": (M, 2048, 512, L.EPI_MUL_DGELU), "dqkv": (M, 512, 1536, L.EPI_STORE), "dfc1": (M, 512, 2048, L.EPI_STORE), "dproj": (M, 512, 512, L.EPI_STORE),
      "sq4k": (4096, 4096, 4096, L.EPI_STORE),
      # backbone 1x1 convolutions at cfg3 (128 frames), forward shapes (M, O, I); the input gradients are the same shapes with O and I swapped
      "s1 64>256": (401408, 256, 64, L.EPI_STORE), "s1 256>64": (401408, 64, 256, L.EPI_STORE), "s2 128>512": (100352, 512, 128, L.EPI_STORE),
      "s2 512>128": (100352, 128, 512, L.EPI_STORE), "s3 256>1024": (25088, 1024, 256, L.EPI_STORE), "s3 1024>256": (25088, 256, 1024, L.EPI_STORE),
      "embed 1024>512": (25088, 512, 1024, L.EPI_STORE)}
# weight gradients dW (N, K) = Y(M,N)^T X(M,K)
TN = {"w qkv": (M, 1536, 512), "w fc1": (M, 2048, 512), "w fc2": (M, 512, 2048), "w proj": (M, 512, 512),
      "w s1 64>256": (401408, 256, 64), "w s1 256>64": (401408, 64, 256), "w s2 128>512": (100352, 512, 128), "w s2 512>128": (100352, 128, 512),
      "w s3 256>1024": (25088, 1024, 256), "w s3 1024>256": (25088, 256, 1024)}


def timeit(fn, n):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


print(f"# torch {torch.__version__}; preferred BLAS backend: {torch.backends.cuda.preferred_blas_library()}", flush=True)
print(f"# {'shape':16s} {'M x N x K':>20s}  {'ours us':>8s} {'TF':>7s}   {'vendor us':>9s} {'TF':>7s}   ours/vendor   vendor + the epilogue ours fuses (cheapest ATen form)  ours/that")
worse = []


def vendor_with_epilogue(epi, A, Bt, bias, aux, vout, out):
    """the vendor GEMM followed by what our fused epilogue also does, in its cheapest framework form (bias through addmm = the vendor's own epilogue): the
    like-for-like comparison for the shapes whose plain-GEMM ratio only says that ours moves 2-4x the bytes"""
    if epi == L.EPI_GELU:                      # pre-activation (kept for GELU') + activation
        h = torch.addmm(bias.to(vout.dtype), A, Bt, out=vout)
        return torch.nn.functional.gelu(h)
    if epi == L.EPI_RESID_F32:                 # fp32 residual stream += bf16 branch output
        h = torch.addmm(bias.to(vout.dtype), A, Bt, out=vout)
        return torch.add(aux, h, out=out)
    if epi == L.EPI_MUL_DGELU:                 # (dy W) * GELU'(pre-activation)
        h = torch.mm(A, Bt, out=vout)
        return torch.ops.aten.gelu_backward(h, aux)
    return torch.mm(A, Bt, out=vout)


for name, (m, n, k, epi) in NT.items():
    A = [torch.randn(m, k, device="cuda").bfloat16() for _ in range(3)]
    B = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16()
    Bt = B.t()
    bias = torch.randn(n, device="cuda") if epi in (L.EPI_GELU, L.EPI_RESID_F32) else None
    aux = torch.randn(m, n, device="cuda") if epi == L.EPI_RESID_F32 else torch.randn(m, n, device="cuda").bfloat16() if epi == L.EPI_MUL_DGELU else None
    out = torch.empty(m, n, device="cuda", dtype=torch.float32 if epi == L.EPI_RESID_F32 else torch.bfloat16)
    out2 = torch.empty_like(out) if epi == L.EPI_GELU else None
    vout = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    t_o = timeit(lambda i: ops.gemm_nt(A[i % 3], B, epi, bias=bias, out=out, out2=out2, aux=aux), iters)
    t_v = timeit(lambda i: torch.mm(A[i % 3], Bt, out=vout), iters)
    fl = 2.0 * m * n * k / 1e6
    r = t_o / t_v
    fused = epi in (L.EPI_GELU, L.EPI_RESID_F32, L.EPI_MUL_DGELU)
    t_ve = timeit(lambda i: vendor_with_epilogue(epi, A[i % 3], Bt, bias, aux, vout, out), iters) if fused else t_v
    if t_o / t_ve > 1.15:
        worse.append(name)
    tail = f"   {t_ve:9.1f} us   {t_o / t_ve:5.2f}" if fused else ""
    print(f"NT {name:16s} {m:7d}x{n:5d}x{k:5d}  {t_o:8.1f} {fl / t_o:7.1f}   {t_v:9.1f} {fl / t_v:7.1f}   {r:5.2f}{tail}", flush=True)
    del A, B, out, vout, aux
for name, (m, n, k) in TN.items():
    Y = [torch.randn(m, n, device="cuda").bfloat16() for _ in range(3)]
    X = [torch.randn(m, k, device="cuda").bfloat16() for _ in range(3)]
    dW = torch.zeros(n, k, device="cuda")
    vout = torch.empty(n, k, device="cuda", dtype=torch.bfloat16)
    t_o = timeit(lambda i: ops.gemm_tn_wgrad(Y[i % 3], X[i % 3], dW=dW), iters)
    t_v = timeit(lambda i: torch.mm(Y[i % 3].t(), X[i % 3], out=vout), iters)
    fl = 2.0 * m * n * k / 1e6
    r = t_o / t_v
    if r > 1.15:
        worse.append(name)
    print(f"TN {name:16s} {m:7d}x{n:5d}x{k:5d}  {t_o:8.1f} {fl / t_o:7.1f}   {t_v:9.1f} {fl / t_v:7.1f}   {r:5.2f}", flush=True)
    del Y, X
print(f"# shapes where the vendor path (GEMM + the epilogue ours fuses, where there is one) is > 1.15x faster than ours: {', '.join(worse) if worse else 'none'}")
