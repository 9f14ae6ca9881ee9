"""Measurement: GroupNorm backward (reduce pass + apply pass, both read dy and x) on a stage-1 activation (128 x 56 x 56 x 256, bf16: 205 MB per tensor, more than
the 256 MB Infinity Cache holds for dy + x) in ONE call vs in frame chunks (reduce(chunk), apply(chunk), ...): does the apply pass of a chunk find the chunk's
dy / x still on-die?  Caches are flushed (1 GB written) before every timed call.   python scripts/gn_chunk_probe.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd import _lib as L, ops  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda")
    lib, p = L.lib(), ops._p
    for (N, HW, C, res) in ((128, 3136, 256, True), (128, 3136, 64, False), (128, 784, 512, True), (128, 196, 1024, True)):
        torch.manual_seed(0)
        x = torch.randn(N, HW, C, device=dev).bfloat16()
        r = torch.randn(N, HW, C, device=dev).bfloat16() if res else None
        dy = torch.randn(N, HW, C, device=dev).bfloat16()
        gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        y = torch.empty_like(x)
        sums = torch.zeros(N, 32, 2, dtype=torch.float64, device=dev)
        mask = torch.empty(N * HW * C // 8, dtype=torch.uint8, device=dev) if res else None
        st = torch.cuda.current_stream().cuda_stream
        ops.check(lib.maed_groupnorm_fwd(p(x), p(r), p(gamma), p(beta), p(y), p(sums), p(mask), N, HW, C, 1e-5, 1, L.BF16, 1, st), "fwd")
        dx = torch.empty_like(x)
        flush = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        results = {}
        for chunk in (N, 64, 32, 16, 8):
            tot = 0.0
            outs = None
            for it in range(iters + 1):
                flush.fill_(it & 1)
                dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
                ab = torch.zeros(N, C, 2, device=dev)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for n0 in range(0, N, chunk):
                    sl = slice(n0, n0 + chunk)
                    mk = mask[n0 * HW * C // 8:(n0 + chunk) * HW * C // 8] if res else None
                    ops.check(lib.maed_groupnorm_bwd(p(x[sl]), p(mk), p(dy[sl]), p(sums[sl]), p(gamma), p(beta), p(dx[sl]), None, p(dg), p(db), p(ab[sl]),
                                                     chunk, HW, C, 1e-5, 1, L.BF16, 1, None, None, st), "bwd")      # (frame_sync = None: the two-pass kernels this probe is about)
                e1.record()
                torch.cuda.synchronize()
                if it:
                    tot += e0.elapsed_time(e1)
                outs = (dx.float().abs().sum().item(), dg.abs().sum().item())
            results[chunk] = (1e3 * tot / iters, outs)
        base = results[N]
        mb = N * HW * C * 2 / 1e6
        print(f"GN bwd N={N} HW={HW} C={C} residual+mask={res} ({mb:.0f} MB per tensor): " +
              "  ".join(f"chunk {c}: {t:7.1f} us" for c, (t, _) in results.items()) +
              f"   checks equal: {all(abs(o[0] - base[1][0]) <= 1e-6 * abs(base[1][0]) for _, o in results.values())}")


if __name__ == "__main__":
    main()
