"""is the 77 MB qkv tensor the spatial / temporal attention forward reads served from the Infinity Cache when the attention runs directly
behind the qkv GEMM (as in the step), or from HBM?  Three conditions per kernel, each preceded by a marker fill so the kernel trace can be split:
  0 "in situ":  qkv GEMM (writes qkv) -> attention                  (what the step does)
  1 "cold":     qkv GEMM -> 1 GB of unrelated stores (evicts L2 and the 256 MB Infinity Cache) -> attention
  2 "hot":      attention twice on the same tensor: the second launch
Run under rocprofv3 --kernel-trace (durations) and, separately, --pmc FETCH_SIZE (on gfx950 FETCH_SIZE counts L2 -> fabric requests, Infinity-Cache hits
included -- MI355X_MICROARCH.md -- so the DURATION is what tells residency; FETCH_SIZE tells over-fetch).  scripts/r6/attn_residency_parse.py reads the traces."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from maed_amd import ops, _lib as L  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
N, T, P, H = 8, 16, 197, 8
C, F_ = 64 * H, 8 * 16
M = F_ * P
torch.manual_seed(0)
x = torch.randn(M, C, device="cuda").bfloat16()
w = (torch.randn(3 * C, C, device="cuda") * C ** -0.5).bfloat16()
qkv = torch.empty(F_, P, 3 * C, device="cuda", dtype=torch.bfloat16)
do = torch.randn(F_, P, C, device="cuda").bfloat16()
junk = torch.empty(256 * 1024 * 1024, device="cuda", dtype=torch.float32)      # 1 GB


def marker(i):
    torch.empty(4096 * (i + 1), device="cuda").fill_(1.0)


def gemm():
    ops.gemm_nt(x, w, L.EPI_STORE, out=qkv.view(M, 3 * C))


for r in range(reps):
    for cond in range(3):
        for kind in range(4):     # 0 spatial fwd, 1 temporal fwd, 2 spatial bwd, 3 temporal bwd
            if kind >= 2:
                o, lse = (ops.attn_spatial_fwd(qkv, H) if kind == 2 else ops.attn_temporal_fwd(qkv, H, T))
            gemm()
            if cond == 1:
                junk.fill_(0.5)
            if cond == 2:
                (ops.attn_spatial_fwd(qkv, H) if kind == 0 else ops.attn_temporal_fwd(qkv, H, T) if kind == 1 else
                 ops.attn_spatial_bwd(qkv, o, do, lse, H) if kind == 2 else ops.attn_temporal_bwd(qkv, o, do, lse, H, T))
            marker(cond * 4 + kind)
            if kind == 0:
                ops.attn_spatial_fwd(qkv, H)
            elif kind == 1:
                ops.attn_temporal_fwd(qkv, H, T)
            elif kind == 2:
                ops.attn_spatial_bwd(qkv, o, do, lse, H)
            else:
                ops.attn_temporal_bwd(qkv, o, do, lse, H, T)
            marker(15)
    torch.cuda.synchronize()
print("done")
