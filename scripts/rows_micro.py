"""row-item 3x3 weight gradient (64 -> 64 channels, 128 x 56 x 56): workgroup-count sweep and the general TN kernel beside it; usage: rows_micro.py [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd import ops, _lib as L
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
x = torch.randn(128, 64, 56, 56, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
dy = torch.randn_like(x)
dW = torch.zeros(64, 3, 3, 64, device="cuda")
def run(): ops.conv3x3_wgrad(dy, x, out=dW)
for wgs in (0, 128, 192, 256, 384, 512):
    L.set_option(L.OPT_CONV3X3_ROWS_WGS, wgs)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    print(f"{'general TN kernel' if wgs == 0 else f'row items, {wgs:3d} workgroups'}: {1e3 * e0.elapsed_time(e1) / iters:7.1f} us", flush=True)
L.set_option(L.OPT_CONV3X3_ROWS_WGS, 256)
