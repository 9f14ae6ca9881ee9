"""Stem convolution (3 -> 64, 7x7, stride 2) at the cfg3 clip (128 frames of 224 x 224): the library's kernels (csrc/stem.hip) against the vendor convolution
F.conv2d dispatches to, same box, same run.  Forward (with and without the GroupNorm statistics epilogue; weights from LDS / in registers), weight gradient
(workgroup-count sweep), and the vendor's forward + weight gradient incl. the framework's own helper kernels (timed as the whole aten call).
usage: stem_micro.py [iters] [frames]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd import ops, _lib as L  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
H = W = 224
dev = "cuda"
torch.manual_seed(0)


def timeit(fn, n=iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


x = torch.randn(N, 3, H, W, device=dev)
w = (torch.randn(64, 3, 7, 7, device=dev) * 147 ** -0.5).bfloat16().contiguous(memory_format=torch.channels_last)
xp = ops.stem_input(x, torch.bfloat16, 7, 2, own=True)
xv = ops.stem_input(x, torch.bfloat16, 7, 2)
dy = torch.randn(N, 64, H // 2, W // 2, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
sums = torch.zeros(N, 32, 2, dtype=torch.float64, device=dev)
dw = torch.zeros(64, 147, device=dev)
wimg = torch.empty(64 * 224, dtype=torch.bfloat16, device=dev)
y = torch.empty(N, 64, H // 2, W // 2, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
lib = L.lib()
wc = w.permute(0, 2, 3, 1)
out_mb = y.numel() * 2 / 1e6
print(f"# {N} frames {H}x{W}: output {out_mb:.0f} MB, padded input {xp.numel() * 2 / 1e6:.0f} MB, 30.2 GFLOP real / 46.0 padded")
print(f"stem_input c_stride 3: {timeit(lambda: ops.stem_input(x, torch.bfloat16, 7, 2)):7.1f} us   c_stride 4: {timeit(lambda: ops.stem_input(x, torch.bfloat16, 7, 2, own=True)):7.1f} us")
for _ in (0,):
    for s_ in (None, sums):
        t = timeit(lambda: ops.check(lib.maed_stem7x7s2_fwd(xp.data_ptr(), wc.data_ptr(), wimg.data_ptr(), y.data_ptr(), None if s_ is None else s_.data_ptr(), N, H, W, L.BF16,
                                                        torch.cuda.current_stream().cuda_stream), "fwd"))
        print(f"own fwd  stats {'yes' if s_ is not None else 'no '}: {t:7.1f} us  ({out_mb / t * 1e3:6.0f} GB/s of output)")
t = timeit(lambda: F.conv2d(xv, w, None, 2, 0))
print(f"vendor fwd (F.conv2d on the padded 3-channel image): {t:7.1f} us")
yv = F.conv2d(xv, w, None, 2, 0)
for wgs, use_sc in ((256, 1), (384, 1), (512, 1), (512, 0), (768, 1), (1024, 1)):
    L.set_option(L.OPT_STEM_WGRAD_WGS, wgs)
    sc = torch.empty(lib.maed_stem7x7s2_wgrad_scratch_floats(N, H, W), device=dev) if use_sc else None
    t = timeit(lambda: ops.check(lib.maed_stem7x7s2_wgrad(dy.data_ptr(), xp.data_ptr(), dw.data_ptr(), sc.data_ptr() if sc is not None else None, N, H, W, L.BF16, torch.cuda.current_stream().cuda_stream), "wgrad"))
    print(f"own wgrad {wgs:5d} workgroups, {'partial slots + reduce' if use_sc else 'atomics'}: {t:7.1f} us")
L.set_option(L.OPT_STEM_WGRAD_WGS, 512)
sc = torch.empty(lib.maed_stem7x7s2_wgrad_scratch_floats(N, H, W), device=dev)
t = timeit(lambda: torch.ops.aten.convolution_backward(dy, xv, w, None, (2, 2), (0, 0), (1, 1), False, (0, 0), 1, (False, True, False)))
print(f"vendor wgrad (aten.convolution_backward, weight only):  {t:7.1f} us")
# agreement
dw.zero_()
ops.check(lib.maed_stem7x7s2_wgrad(dy.data_ptr(), xp.data_ptr(), dw.data_ptr(), sc.data_ptr() if sc is not None else None, N, H, W, L.BF16, torch.cuda.current_stream().cuda_stream), "wgrad")
gv = torch.ops.aten.convolution_backward(dy, xv, w, None, (2, 2), (0, 0), (1, 1), False, (0, 0), 1, (False, True, False))[1].float()
go = dw.view(64, 7, 7, 3).permute(0, 3, 1, 2)
ops.check(lib.maed_stem7x7s2_fwd(xp.data_ptr(), wc.data_ptr(), wimg.data_ptr(), y.data_ptr(), None, N, H, W, L.BF16, torch.cuda.current_stream().cuda_stream), "fwd")
print(f"agreement with the vendor: fwd max |d| {float((y.float() - yv.float()).abs().max()):.3e} (max |y| {float(yv.float().abs().max()):.2f}), "
      f"wgrad rel {float((go - gv).abs().max() / gv.abs().max()):.3e}")
