"""SMPL LBS forward (maed_smpl_lbs_fwd: chain + skinning) and the decoder tail's backward at 128 frames; usage: lbs_micro.py [iters]
(frames per workgroup: MAED_LBS_FB = 4 / 8 / 16)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MAED_SYNTHETIC_SMPL_OK", "1")
from maed_amd.smpl import SMPL
from maed_amd.geometry import rot6d_to_rotmat
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
smpl = SMPL().cuda()
Fr = 128
betas = torch.randn(Fr, 10, device="cuda")
rot = rot6d_to_rotmat(torch.randn(Fr * 24, 6, device="cuda")).reshape(Fr, 24, 3, 3).contiguous()
def run(): smpl.lbs_hip(betas, rot)
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
print(f"maed_smpl_lbs_fwd (chain + skin), 128 frames: {1e3 * e0.elapsed_time(e1) / iters:7.1f} us")
