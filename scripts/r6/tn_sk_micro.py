"""Round 6: the persistent K-stream weight-gradient GEMM (csrc/gemm_tn_sk.hip) at the TN shapes of the benchmarked step -- correctness against fp64, bit-for-bit
determinism over repetitions (no atomics: a fixed reduction order), time against the split-M kernels with closing atomics (MAED_OPT_TN_SK = 0: gemm_tn2.hip) and the
vendor library, interleaved, rotating operands.   usage: tn_sk_micro.py [iters] [reps]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from maed_amd import ops, _lib as L  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
M = 128 * 197
SHAPES = {"w qkv": (M, 1536, 512, True), "w fc1": (M, 2048, 512, True), "w fc2": (M, 512, 2048, True), "w proj": (M, 512, 512, True), "w ts (no bias)": (M, 512, 512, False),
          "w s3 256>1024": (25088, 1024, 256, False), "w s3 1024>256": (25088, 256, 1024, False), "w s3 512>1024 s2": (25088, 1024, 512, False),
          "w s2 128>512": (100352, 512, 128, False), "w embed 1024>512": (25088, 512, 1024, True),
          "cfg5 w qkv": (32896, 2304, 768, True), "cfg5 w fc2": (32896, 768, 3072, True)}
lib = L.lib()


def ev_time(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


print(f"# {'shape':18s} {'M x N x K':>20s} | split-M + atomics (TN_SK=0) | persistent K-stream + reduce (TN_SK=1) | vendor bare   [us median of 5 rounds; TF]")
for name, (m, n, k, bias) in SHAPES.items():
    torch.manual_seed(0)
    Y = [torch.randn(m, n, device="cuda").bfloat16() for _ in range(3)]
    X = [torch.randn(m, k, device="cuda").bfloat16() for _ in range(3)]
    dW = torch.zeros(n, k, device="cuda")
    db = torch.zeros(n, device="cuda") if bias else None
    vout = torch.empty(n, k, device="cuda", dtype=torch.bfloat16)
    ref = (Y[0].double().t() @ X[0].double())
    refb = Y[0].double().sum(0)
    res = {}
    for mode in (0, 1):
        lib.maed_set_option(L.OPT_TN_SK, mode)
        lib.maed_set_option(L.OPT_SK_GRID, 255 if mode else 0)      # (an explicit grid takes the persistent kernel whatever the launcher's heuristic says)
        first, bad = None, 0
        for r in range(reps if mode == 1 else 3):
            d = torch.zeros(n, k, device="cuda")
            b = torch.zeros(n, device="cuda") if bias else None
            ops.gemm_tn_wgrad(Y[0], X[0], dW=d, dbias=b)
            if first is None:
                first = (d, b)
            else:
                bad += int(not torch.equal(d, first[0]) or (bias and not torch.equal(b, first[1])))
        err = (first[0].double() - ref).abs().max().item() / ref.abs().max().item()
        errb = ((first[1].double() - refb).abs().max().item() / refb.abs().max().item()) if bias else 0.0
        res[mode] = (err, errb, bad)
    ts = {"0": [], "1": [], "v": []}
    for rnd_ in range(5):
        for key in ("0", "1"):
            lib.maed_set_option(L.OPT_TN_SK, int(key))
            lib.maed_set_option(L.OPT_SK_GRID, 255 if key == "1" else 0)
            for i in range(2):
                ops.gemm_tn_wgrad(Y[i % 3], X[i % 3], dW=dW, dbias=db)
            ts[key].append(ev_time(lambda i: ops.gemm_tn_wgrad(Y[i % 3], X[i % 3], dW=dW, dbias=db), iters))
        for i in range(2):
            torch.mm(Y[i % 3].t(), X[i % 3], out=vout)
        ts["v"].append(ev_time(lambda i: torch.mm(Y[i % 3].t(), X[i % 3], out=vout), iters))
    lib.maed_set_option(L.OPT_TN_SK, 1)
    lib.maed_set_option(L.OPT_SK_GRID, 0)
    fl = 2.0 * m * n * k / 1e6
    med = {k_: statistics.median(v) for k_, v in ts.items()}
    ok = res[1][0] < 1e-5 and res[1][1] < 1e-5 and res[1][2] == 0
    eligible = n >= 256 and k >= 256 and m % 128 == 0
    print(f"{name:18s} {m:7d}x{n:5d}x{k:5d} | {med['0']:7.1f} {fl / med['0']:7.1f} | {med['1']:7.1f} {fl / med['1']:7.1f} | {med['v']:7.1f} {fl / med['v']:7.1f} |  "
          f"{'ok' if ok else 'WRONG' if eligible else 'n/a (N or K < 256: the split-M kernel in both columns)'}: rel err dW {res[1][0]:.1e} (atomics kernel {res[0][0]:.1e}) dbias {res[1][1]:.1e}; differing reps {res[1][2]}/{reps - 1} (atomics kernel {res[0][2]}/2)", flush=True)
    del Y, X
