"""Golden fixture for the long-clip configuration (BASELINE.json configs[4]: T = 64 frames of 256x256, P = 257 tokens, STE depth 12 / dim 768 / 12 heads,
max_seqlen = 64) at FULL size: the fp32 CPU oracle (oracle/maed_ref.py -- the pinned restatement of lib/models/{resnetv2,vision_transformer,ktd,smpl,spin}.py) on ONE
seeded clip with seeded random-init parameters.  Minutes of CPU work, so it runs HERE and the outputs travel as data: tests/golden/g15_cfg5_full.npz holds theta,
kp_3d, kp_2d, rotmat, a 1-in-53 sample of the vertices and the encoder feature -- inputs and parameters are regenerated from the seeds on the GPU box
(torch's CPU generator is the same bits on both machines: same image).  What it pins that the kernel-level tests cannot: the pos_embed / temp_embed slices at
P = 257 / T = 64 and the whole depth-12 stack wired at full size (VERDICT r3 "What's weak" #9).

    python oracle/make_golden_cfg5.py        # -> tests/golden/g15_cfg5_full.npz  (test infrastructure only)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import maed_ref as R  # noqa: E402

CFG5 = dict(depth=12, H=12, C=768, img=256, T=64, hidden=1024, P=257, param_seed=11, clip_seed=31)


def inputs():
    params = R.make_params(embed_dim=CFG5["C"], depth=CFG5["depth"], hidden_dim=CFG5["hidden"], n_tokens=CFG5["P"], max_seqlen=CFG5["T"], seed=CFG5["param_seed"])
    clip = torch.randn(1, CFG5["T"], 3, CFG5["img"], CFG5["img"], generator=torch.Generator().manual_seed(CFG5["clip_seed"]))
    return params, clip


if __name__ == "__main__":
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    params, clip = inputs()
    t0 = time.time()
    with torch.no_grad():
        feat = R.ste_forward_features(clip.reshape(-1, 3, CFG5["img"], CFG5["img"]), params, "encoder.", CFG5["depth"], CFG5["H"], CFG5["T"])
        out = R.maed_forward(clip, params, R.make_synthetic_smpl(0), depth=CFG5["depth"], H=CFG5["H"])
    print(f"oracle fp32 forward (twice the encoder): {time.time() - t0:.0f} s")
    dst = os.path.join(ROOT, "tests", "golden", "g15_cfg5_full.npz")
    np.savez_compressed(dst, theta=out["theta"].numpy(), kp_3d=out["kp_3d"].numpy(), kp_2d=out["kp_2d"].numpy(), rotmat=out["rotmat"].numpy(),
                        verts_sample=out["verts"][:, :, ::53].numpy(), feature=feat.numpy(),
                        checks=np.array([params["encoder.pos_embed"].double().sum().item(), params["encoder.temp_embed"].double().sum().item(), clip.double().sum().item()]))
    print("wrote", dst, os.path.getsize(dst) // 1024, "KiB")
