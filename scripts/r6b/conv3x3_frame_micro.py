"""Round 6 (second session): the stage-3 3x3 convolutions (forward with GroupNorm statistics, forward plain, stride-1 input gradient) on one frame x 128 channels per
workgroup with a three-stage copy ring (conv3x3_frame_bf16_kernel, MAED_OPT_CONV3X3_FRAME = 1) against the 128 x 128 tiles (= 0), interleaved, bit-compared;
cfg3 (14 x 14, 256 channels, 128 frames) and cfg5 (16 x 16, 256 channels, 128 frames).   usage: conv3x3_frame_micro.py [iters]"""
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


for H, C in [(14, 256), (16, 256), (14, 512)]:
    x = torch.randn(Fr, C, H, H, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, C, 3, 3, device="cuda") * (9 * C) ** -0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    w_taps = w.permute(0, 2, 3, 1)
    wimg = w_taps.permute(1, 2, 3, 0).contiguous()
    cases = {"fwd + GN stats": lambda: ops.conv3x3(x, w_taps, 1, gn_sums=torch.zeros(Fr, 32, 2, dtype=torch.float64, device="cuda")),
             "fwd": lambda: ops.conv3x3(x, w_taps, 1), "dgrad": lambda: ops.conv3x3(x, wimg, 1, w_layout=1)}
    for name, fn in cases.items():
        res, outs = {0: [], 1: []}, {}
        for rnd in range(5):
            for mode in (0, 1):
                lib.maed_set_option(L.OPT_CONV3X3_FRAME, mode)
                outs[mode] = fn()
                res[mode].append(ev(fn))
        lib.maed_set_option(L.OPT_CONV3X3_FRAME, 1)
        gf = 2.0 * Fr * H * H * C * 9 * C
        t0, t1 = statistics.median(res[0]), statistics.median(res[1])
        print(f"H={H:3d} C={C:4d} {name:15s}: 128 x 128 tiles {t0:6.1f} us ({gf / t0 / 1e6:5.0f} TF)   one frame per workgroup {t1:6.1f} us ({gf / t1 / 1e6:5.0f} TF)   "
              f"bit-equal: {torch.equal(outs[0], outs[1])}", flush=True)
