"""Round 6: the stage-2 / stage-3 3x3 convolutions (forward and stride-1 input gradient) on 128 x 128 tiles (default) against 128 x 64 tiles
(MAED_OPT_CONV3X3_NARROW_WGS: twice the workgroups where the grid is small), interleaved, bit-compared.   usage: conv3x3_narrow_micro.py [iters]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from maed_amd import ops, _lib as L  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib = L.lib()
Fr = 128


def ev(fn):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


for H, C, cnt in [(56, 64, 3), (28, 128, 3), (14, 256, 8)]:
    x = torch.randn(Fr, C, H, H, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, C, 3, 3, device="cuda") * (9 * C) ** -0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    w_taps = w.permute(0, 2, 3, 1)
    res, outs = {}, {}
    for thr in (0, 600, 2048):
        res[thr] = []
    for rnd in range(5):
        for thr in (0, 600, 2048):
            lib.maed_set_option(L.OPT_CONV3X3_NARROW_WGS, thr)
            outs[thr] = ops.conv3x3(x, w_taps, 1)
            res[thr].append(ev(lambda: ops.conv3x3(x, w_taps, 1)))
    lib.maed_set_option(L.OPT_CONV3X3_NARROW_WGS, 0)
    tiles = (Fr * H * H + 127) // 128 * ((C + 127) // 128)
    print(f"H={H:3d} C={C:4d} (x{cnt} per pass, {tiles} workgroups of 128x128): " + "  ".join(f"narrow<{thr}: {statistics.median(v):6.1f} us" for thr, v in res.items())
          + f"   bit-equal: {torch.equal(outs[0], outs[600]) and torch.equal(outs[0], outs[2048])}", flush=True)
