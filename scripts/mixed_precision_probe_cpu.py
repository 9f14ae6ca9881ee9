"""CPU estimate (oracle arithmetic, no GPU) of what a MIXED mode would give at full cfg3 module size (one 16-frame 224^2 clip): the backbone in fp32, everything behind it
(1x1 projection, STE blocks, pre_logits, KTD head) with bf16 matrix products (torch.autocast on the oracle: bf16 GEMM operands, fp32 LayerNorm / softmax / residual
stream -- the arithmetic of the product's bf16 mode) -- theta / kp_3d error against the all-fp32 oracle.  Decides whether the parity-at-speed mode of round 4 may keep
the STE in bf16.   python scripts/mixed_precision_probe_cpu.py"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import maed_ref as R  # noqa: E402

depth, H, img, hidden, T = 6, 8, 224, 1024, 16
C, P = 64 * H, (img // 16) ** 2 + 1
torch.set_num_threads(min(32, os.cpu_count() or 1))
for seed in (7, 8):
    params = R.make_params(embed_dim=C, depth=depth, hidden_dim=hidden, n_tokens=P, seed=seed)
    sp = R.make_synthetic_smpl(0)
    clip = torch.randn(1, T, 3, img, img, generator=torch.Generator().manual_seed(21 + seed))
    x = clip.reshape(-1, 3, img, img)
    t0 = time.time()
    with torch.no_grad():
        feat = R.resnetv2_features(x, params, "encoder.patch_embed.backbone.")

        def tail(feat, bf16):
            with torch.autocast("cpu", dtype=torch.bfloat16, enabled=bf16):
                p, pre = params, "encoder."
                tok = F.conv2d(feat, p[pre + "patch_embed.proj.weight"], p[pre + "patch_embed.proj.bias"]).flatten(2).transpose(1, 2)
                xx = R.embed_tokens(tok.float(), p, pre, T)
                for i in range(depth):
                    xx = R.block(xx, p, f"{pre}blocks.{i}.", H, T)
                xx = R.layer_norm(xx.float(), p[pre + "norm.weight"], p[pre + "norm.bias"])[:, 0]
                xf = torch.tanh(F.linear(xx, p[pre + "pre_logits.fc.weight"], p[pre + "pre_logits.fc.bias"]))
                pose, shape, cam = R.ktd_head(xf, p, "decoder.")
            return R.ktd_get_output(pose.float(), shape.float(), cam.float(), sp), xf.float()
        o32, f32 = tail(feat, False)
        o16, f16 = tail(feat, True)
        ob, fb = tail(feat.bfloat16().float(), True)          # + the backbone's OUTPUT rounded to bf16 (what the projection GEMM would read)
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
    print(f"seed {seed}: ({time.time() - t0:.0f}s)  feature {rel(f16, f32):.2e}   theta {rel(o16['theta'], o32['theta']):.2e}   kp_3d {rel(o16['kp_3d'], o32['kp_3d']):.2e}   "
          f"kp_2d {rel(o16['kp_2d'], o32['kp_2d']):.2e}   verts {rel(o16['verts'], o32['verts']):.2e}   | with bf16-rounded backbone output: theta {rel(ob['theta'], o32['theta']):.2e}", flush=True)
