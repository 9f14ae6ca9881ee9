"""Round 6: the persistent K-stream GEMM (csrc/gemm_sk.hip) at the NT shapes of the benchmarked step -- correctness against an fp64 product, run-to-run
determinism over many repetitions (a race in the copy ring or a stale slab shows up as a differing repetition), and time against the per-tile kernels of
rounds 1-5 (MAED_OPT_SK = 0) and the vendor library's bare GEMM, interleaved in one process, rotating operands.

usage: sk_micro.py [iters] [reps] [shape-name-filter]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from maed_amd import ops, _lib as L  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
filt = sys.argv[3] if len(sys.argv) > 3 else ""
M = 128 * 197
SHAPES = {"qkv": (M, 1536, 512, L.EPI_STORE), "fc1+gelu": (M, 2048, 512, L.EPI_GELU), "fc2+res": (M, 512, 2048, L.EPI_RESID_F32), "proj+res": (M, 512, 512, L.EPI_RESID_F32),
          "dfc2*gelu'": (M, 2048, 512, L.EPI_MUL_DGELU), "dqkv": (M, 512, 1536, L.EPI_STORE), "dfc1": (M, 512, 2048, L.EPI_STORE), "dproj": (M, 512, 512, L.EPI_STORE),
          "sq4k": (4096, 4096, 4096, L.EPI_STORE), "s3 1024>256": (25088, 256, 1024, L.EPI_STORE), "embed 1024>512": (25088, 512, 1024, L.EPI_STORE),
          "cfg5 qkv": (32896, 2304, 768, L.EPI_STORE), "cfg5 fc2": (32896, 768, 3072, L.EPI_RESID_F32)}
lib = L.lib()


def ev_time(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


print(f"# {'shape':16s} {'M x N x K':>20s} | per-tile kernels (SK=0) | persistent, no K cuts (SK=2) | persistent + stream-K (SK=3) | heuristic (SK=1) | vendor bare   [us median of 5 rounds; TF]")
for name, (m, n, k, epi) in SHAPES.items():
    if filt and filt not in name:
        continue
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

    def run(i, impl=L.IMPL_AUTO, o=out, o2=out2):
        return ops.gemm_nt(A[i % 3], B, epi, bias=bias, out=o, out2=o2, aux=aux, impl=impl)

    # ---- correctness: rows sampled against fp64 (all tiles in M: every 37th row + the last rows), every column
    rows = torch.cat([torch.arange(0, m, 37, device="cuda"), torch.arange(max(0, m - 300), m, device="cuda")]).unique()
    acc = (A[0][rows].double() @ B.double().t())
    if epi == L.EPI_STORE:
        want = acc + bias.double()
    elif epi == L.EPI_GELU:
        pre = (acc + bias.double()).bfloat16().double()
        want = torch.nn.functional.gelu(pre)
    elif epi == L.EPI_RESID_F32:
        want = aux[rows].double() + acc + bias.double()
    else:
        x = aux[rows].double()
        want = acc * (0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5)
    res = {}
    for mode in (2, 3):
        lib.maed_set_option(L.OPT_SK, mode)
        o = torch.full_like(out, float("nan"))
        o2 = torch.full_like(out, float("nan")) if out2 is not None else None
        run(0, L.IMPL_MFMA_SK, o, o2)
        torch.cuda.synchronize()
        err = (o[rows].double() - want).abs().max().item()
        tol = (2 ** -7 if odt == torch.bfloat16 else 1e-4) * want.abs().max().item() + 1e-5
        nan = int(torch.isnan(o.float()).sum().item())
        # determinism: the same launch `reps` times into fresh poisoned outputs
        bad = 0
        for _ in range(reps):
            o3 = torch.full_like(out, float("nan"))
            run(0, L.IMPL_MFMA_SK, o3, o2)
            bad += int(not torch.equal(o3, o))
        res[mode] = (err, tol, nan, bad)
    ok = all(e <= t and nn_ == 0 and b == 0 for (e, t, nn_, b) in res.values())
    # ---- time: 5 interleaved rounds
    ts = {"0": [], "2": [], "3": [], "1": [], "v": []}
    for rnd_ in range(5):
        for key in ("0", "2", "3", "1"):
            lib.maed_set_option(L.OPT_SK, int(key))
            impl = L.IMPL_AUTO if key in ("0", "1") else L.IMPL_MFMA_SK
            for i in range(2):
                run(i, impl)
            ts[key].append(ev_time(lambda i: run(i, impl), iters))
        for i in range(2):
            torch.mm(A[i % 3], Bt, out=vout)
        ts["v"].append(ev_time(lambda i: torch.mm(A[i % 3], Bt, out=vout), iters))
    lib.maed_set_option(L.OPT_SK, 1)
    fl = 2.0 * m * n * k / 1e6
    med = {k_: statistics.median(v) for k_, v in ts.items()}
    line = f"{name:16s} {m:7d}x{n:5d}x{k:5d} |"
    for key in ("0", "2", "3", "1", "v"):
        line += f" {med[key]:7.1f} {fl / med[key]:7.1f} |"
    line += f"  {'ok' if ok else 'WRONG'}: " + " ".join(f"SK={mo} err {e:.2e}/{t:.1e} nan {nn_} differing reps {b}/{reps}" for mo, (e, t, nn_, b) in res.items())
    print(line, flush=True)
    del A, B, out, vout, aux
print("faults:", lib.maed_device_faults())
