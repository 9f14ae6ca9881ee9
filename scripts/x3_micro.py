"""The split-bf16 fp32 kernels (csrc/gemm_x3.hip, attn_x3.hip) in isolation at the cfg3 shapes, next to their bf16 twins: NT GEMMs of one STE block,
weight-gradient (TN) GEMMs of the STE and of the backbone's 1x1 / 3x3 convolutions, 3x3 implicit GEMM forward, spatial attention forward / backward.
usage: x3_micro.py [iters] [groups: nt,tn,conv,attn]   (event timing on the current stream, rotating operands)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maed_amd  # noqa: E402
from maed_amd import ops, _lib as L  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
groups = (sys.argv[2] if len(sys.argv) > 2 else "nt,tn,conv,attn").split(",")
only = sys.argv[3].split(",") if len(sys.argv) > 3 and sys.argv[3] != "all" else None      # shape-name filter (nt / tn groups)
only_modes = sys.argv[4].split(",") if len(sys.argv) > 4 else None                         # mode filter: bf16,bf16x3,bf16x6
torch.manual_seed(0)
M = 128 * 197
dev = "cuda"


def timeit(fn, n=iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def modes():
    if only_modes is None or "bf16" in only_modes:
        yield "bf16", torch.bfloat16
    for m in ("bf16x3", "bf16x6"):
        if only_modes is None or m in only_modes:
            maed_amd.set_float32_matmul_precision(m)
            yield m, torch.float32
    maed_amd.set_float32_matmul_precision("exact")


if "nt" in groups:
    shapes = {"qkv": (M, 1536, 512, L.EPI_STORE), "fc1": (M, 2048, 512, L.EPI_GELU), "fc2": (M, 512, 2048, L.EPI_RESID_F32), "proj": (M, 512, 512, L.EPI_RESID_F32),
              "dfc2": (M, 2048, 512, L.EPI_MUL_DGELU), "dqkv": (M, 512, 1536, L.EPI_STORE), "dfc1": (M, 512, 2048, L.EPI_STORE), "dproj": (M, 512, 512, L.EPI_STORE),
              "sq4k": (4096, 4096, 4096, L.EPI_STORE), "c3a": (25088, 256, 1024, L.EPI_STORE), "c3b": (25088, 1024, 256, L.EPI_STORE),
              "c2a": (100352, 128, 512, L.EPI_STORE), "c2b": (100352, 512, 128, L.EPI_STORE), "c1a": (401408, 64, 256, L.EPI_STORE), "c1b": (401408, 256, 64, L.EPI_STORE)}
    for name, (m, n, k, epi) in shapes.items():
        if only and name not in only:
            continue
        line = f"nt {name:5s} {m}x{n}x{k}:"
        for mode, dt in modes():
            A = [torch.randn(m, k, device=dev).to(dt) for _ in range(2)]
            B = (torch.randn(n, k, device=dev) * k ** -0.5).to(dt)
            bias = torch.randn(n, device=dev) if epi != L.EPI_MUL_DGELU else None
            aux = torch.randn(m, n, device=dev) if epi == L.EPI_RESID_F32 else torch.randn(m, n, device=dev).to(dt) if epi == L.EPI_MUL_DGELU else None
            out = torch.empty(m, n, device=dev, dtype=torch.float32 if epi == L.EPI_RESID_F32 else dt)
            out2 = torch.empty_like(out) if epi == L.EPI_GELU else None
            i = [0]
            def fn():
                i[0] += 1
                ops.gemm_nt(A[i[0] & 1], B, epi, bias=bias, out=out, out2=out2, aux=aux)
            us = timeit(fn)
            line += f"  {mode} {us:7.1f} us ({2.0 * m * n * k / us / 1e6:6.0f} TF)"
            del A, B, out, out2, aux
        print(line, flush=True)

if "tn" in groups:
    ste = [("qkv", M, 1536, 512), ("fc1", M, 2048, 512), ("fc2", M, 512, 2048), ("proj", M, 512, 512),
           ("c1 56 64->256", 401408, 256, 64), ("c1 56 256->64", 401408, 64, 256), ("c2 28 128->512", 100352, 512, 128), ("c2 28 512->128", 100352, 128, 512),
           ("c3 14 256->1024", 25088, 1024, 256), ("c3 14 1024->256", 25088, 256, 1024)]
    for name, m, n, k in ste:
        if only and name.split()[0] not in only:
            continue
        line = f"tn {name:16s} M={m} N={n} K={k}:"
        for mode, dt in modes():
            Y, X = torch.randn(m, n, device=dev).to(dt), torch.randn(m, k, device=dev).to(dt)
            dW, db = torch.zeros(n, k, device=dev), torch.zeros(n, device=dev)
            us = timeit(lambda: ops.gemm_tn_wgrad(Y, X, dW=dW, dbias=db))
            line += f"  {mode} {us:7.1f} us ({2.0 * m * n * k / us / 1e6:6.0f} TF)"
            del Y, X
        print(line, flush=True)

if "conv" in groups:
    Fr = 128
    for H, C_, s in ((56, 64, 1), (56, 128, 2), (28, 128, 1), (28, 256, 2), (14, 256, 1)):
        line = f"conv3x3 {H}x{H} C={C_} s{s}:"
        for mode, dt in modes():
            x = torch.randn(Fr, C_, H, H, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
            w = (torch.randn(C_, 3, 3, C_, device=dev) * (9 * C_) ** -0.5).to(dt)
            us = timeit(lambda: ops.conv3x3(x, w, s))
            Ho = -(-H // s)
            fl = 2.0 * Fr * Ho * Ho * C_ * 9 * C_
            line += f"  {mode} fwd {us:7.1f} us ({fl / us / 1e6:5.0f} TF)"
            if s == 1:
                dy = torch.randn(Fr, C_, H, H, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
                dW = torch.zeros(C_, 3, 3, C_, device=dev)
                us = timeit(lambda: ops.conv3x3_wgrad(dy, x, out=dW))
                line += f" wgrad {us:7.1f} us ({fl / us / 1e6:5.0f} TF)"
                del dy
            del x, w
        print(line, flush=True)

if "attn" in groups:
    for P, H, Fr in ((197, 8, 128), (257, 12, 128)):
        line = f"attn spatial P={P} H={H} F={Fr}:"
        for mode, dt in modes():
            qkv = torch.randn(Fr, P, 3 * 64 * H, device=dev).to(dt)
            do = torch.randn(Fr, P, 64 * H, device=dev).to(dt)
            o, lse = ops.attn_spatial_fwd(qkv, H)
            dq = torch.empty_like(qkv)
            uf = timeit(lambda: ops.attn_spatial_fwd(qkv, H))
            ub = timeit(lambda: ops.attn_spatial_bwd(qkv, o, do, lse, H, dqkv=dq))
            fl = 4.0 * P * P * 64 * H * Fr
            line += f"  {mode} fwd {uf:7.1f} us ({fl / uf / 1e6:5.0f} TF) bwd {ub:7.1f} us"
            del qkv, do, o, dq
        print(line, flush=True)
    for P, H, N, T in ((197, 8, 8, 16),):
        line = f"attn temporal P={P} H={H} N={N} T={T}:"
        for mode, dt in modes():
            qkv = torch.randn(N * T, P, 3 * 64 * H, device=dev).to(dt)
            do = torch.randn(N * T, P, 64 * H, device=dev).to(dt)
            o, lse = ops.attn_temporal_fwd(qkv, H, T)
            dq = torch.empty_like(qkv)
            uf = timeit(lambda: ops.attn_temporal_fwd(qkv, H, T))
            ub = timeit(lambda: ops.attn_temporal_bwd(qkv, o, do, lse, H, T, dqkv=dq))
            gb = (4 + 8) * qkv.numel() / 3 * qkv.element_size() / 1e3          # fwd: q,k,v + o; bwd: q,k,v,o,dO + dq,dk,dv
            line += f"  {mode} fwd {uf:7.1f} us bwd {ub:7.1f} us ({gb / (uf + ub):5.0f} GB/s algorithmic)"
            del qkv, do, o, dq
        print(line, flush=True)

