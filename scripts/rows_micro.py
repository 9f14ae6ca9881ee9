"""row-item 3x3 weight gradient (64 -> 64 channels, 128 x 56 x 56) timing; env MAED_R3_DBG ablation bits (diagnostic builds only)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd import ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
x = torch.randn(128, 64, 56, 56, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
dy = torch.randn_like(x)
dW = torch.zeros(64, 3, 3, 64, device="cuda")
def run(): ops.conv3x3_wgrad(dy, x, out=dW)
for tag in sys.argv[2:] or ["0"]:
    k, v = (tag.split("=") + ["0"])[:2] if "=" in tag else ("MAED_R3_DBG", tag)
    os.environ[k] = v
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    print(f"{k}={v}: {1e3 * e0.elapsed_time(e1) / iters:7.1f} us", flush=True)
