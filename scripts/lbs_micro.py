"""SMPL LBS forward (maed_smpl_lbs_fwd: chain + skinning kernels) at 128 frames; MAED_HIP_LIB selects a build variant (frames per workgroup)"""
import os, sys, torch
os.environ.setdefault("MAED_SYNTHETIC_SMPL_OK", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd.smpl import SMPL
smpl = SMPL().cuda()
F = 128
g = torch.Generator().manual_seed(0)
betas = torch.randn(F, 10, generator=g).cuda() * 0.5
rot = torch.linalg.qr(torch.randn(F, 24, 3, 3, generator=g))[0].cuda()
v0, j0 = smpl.lbs_hip(betas, rot)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    v, j = smpl.lbs_hip(betas, rot)
e1.record(); torch.cuda.synchronize()
print(f"lbs forward, {F} frames: {1e3 * e0.elapsed_time(e1) / 50:.1f} us per call; checksum {v.double().sum().item():.6f} {j.double().sum().item():.6f}")
