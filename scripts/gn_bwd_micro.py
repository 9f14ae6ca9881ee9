"""GroupNorm backward at the backbone's cfg3 layer shapes (128 frames, bf16): the two-pass kernels (reduction pass + apply pass: x and dy read twice) against the
one-pass kernel of round 4 (register-resident slices + per-frame arrival counter: 3 tensor streams instead of 5), same box, same run, rotating operands.
Prints per shape: two-pass us, one-pass us, algorithmic GB/s of the one-pass kernel (x + dy + dx [+ mask]), max |dx difference| relative to max |dx|.
usage: gn_bwd_micro.py [iters] [f32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd import ops, _lib as L  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dtype = torch.float32 if "f32" in sys.argv[2:] else torch.bfloat16
N = 128
# (HW, C, relu, residual-mask, layers of that shape in the backbone)
SHAPES = [(112 * 112, 64, 1, 0, 1), (56 * 56, 64, 1, 0, 6), (56 * 56, 256, 1, 1, 3), (56 * 56, 256, 0, 0, 1), (56 * 56, 128, 1, 0, 1), (28 * 28, 128, 1, 0, 7),
          (28 * 28, 512, 1, 1, 4), (28 * 28, 512, 0, 0, 1), (28 * 28, 256, 1, 0, 1), (14 * 14, 256, 1, 0, 17), (14 * 14, 1024, 1, 1, 9), (14 * 14, 1024, 0, 0, 1)]
p = lambda t: None if t is None else t.data_ptr()
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
MODES = (0, 1, 3, 10, 14, 16)        # 3: DIAGNOSTIC -- the one-pass kernel without its frame barrier (wrong results: what the barrier + lockstep cost)
tot = {m: 0.0 for m in MODES}
for HW, C, relu, res, cnt in SHAPES:
    torch.manual_seed(0)
    es = 2 if dtype == torch.bfloat16 else 4
    xs = [torch.randn(N, HW, C, device="cuda").to(dtype) for _ in range(3)]
    dys = [torch.randn(N, HW, C, device="cuda").to(dtype) for _ in range(3)]
    gamma, beta = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    mask = torch.randint(0, 256, (N * HW * C // 8,), dtype=torch.uint8, device="cuda") if res else None
    sums = torch.zeros(N, 32, 2, dtype=torch.float64, device="cuda")
    for g in range(32):     # statistics of x[0] (every x[i] has the same distribution; the kernels only need SOME consistent statistics)
        v = xs[0].float().view(N, HW, 32, C // 32)[:, :, g, :].double()
        sums[:, g, 0] = v.sum((1, 2)); sums[:, g, 1] = (v * v).sum((1, 2))
    dx = torch.empty_like(xs[0])
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    res_out = {}
    t = {}
    for mode in MODES:
        L.set_option(L.OPT_GN_BWD_ONEPASS, mode)

        def run(i):
            ab = torch.zeros(N * C * 2 + N * ops.GN_SYNC_WORDS, device="cuda")
            ops.check(lib.maed_groupnorm_bwd(p(xs[i % 3]), p(mask), p(dys[i % 3]), p(sums), p(gamma), p(beta), p(dx), None, p(dg), p(db), p(ab), N, HW, C, 1e-5, relu,
                                             ops.dt_code(dtype), 1, p(ab) + 4 * N * C * 2, None, st), "bwd")
        for i in range(3):
            run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            run(i)
        e1.record(); torch.cuda.synchronize()
        t[mode] = 1e3 * e0.elapsed_time(e1) / iters
        run(0); torch.cuda.synchronize()
        res_out[mode] = dx.float().clone()
        tot[mode] += cnt * t[mode]
    L.set_option(L.OPT_GN_BWD_ONEPASS, 1)
    err = ((res_out[1] - res_out[0]).abs().max() / res_out[0].abs().max()).item()
    nbytes = N * HW * C * es * 3 + (N * HW * C // 8 if res else 0)
    print(f"HW={HW:6d} C={C:5d} relu={relu} mask={res} x{cnt:2d}: two-pass {t[0]:7.1f} us   one-pass {t[1]:7.1f} us ({nbytes / t[1] / 1e6:5.2f} TB/s algorithmic)   no-barrier (diagnostic) {t[3]:7.1f} us   skew 0/4/6 {t[10]:6.1f} {t[14]:6.1f} {t[16]:6.1f} (default: 2 units)   "
          f"max|d dx| / max|dx| = {err:.2e}  finite={bool(torch.isfinite(res_out[1]).all())}", flush=True)
print(f"backbone total per step (incl. the ab zero-fill and the dgamma/dbeta column sum of each call): two-pass {tot[0] / 1e3:.3f} ms   one-pass {tot[1] / 1e3:.3f} ms   no-barrier {tot[3] / 1e3:.3f} ms")
