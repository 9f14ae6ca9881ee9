"""bf16 vs f32 error of the hybrid R50 backbone, layer by layer (module forward hooks), at the cfg3 input size.
usage: diag_backbone_bf16.py [n_frames=4] [img=224]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd.resnetv2 import ResNetV2
from oracle import maed_ref as R
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 4
img = int(sys.argv[2]) if len(sys.argv) > 2 else 224
dev = torch.device("cuda", 0)
params = R.make_params(embed_dim=512, depth=1, hidden_dim=64, n_tokens=(img // 16) ** 2 + 1, seed=7)
pre = "encoder.patch_embed.backbone."
sd = {k[len(pre):]: v for k, v in params.items() if k.startswith(pre)}
x = torch.randn(nf, 3, img, img, generator=torch.Generator().manual_seed(21))
recs = {}
for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
    m = ResNetV2(layers=(3, 4, 9), compute_dtype=dt)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    rec = {}
    hooks = []
    for n, mod in m.named_modules():
        if n and (n.count(".") <= 3) and not n.endswith("downsample"):
            hooks.append(mod.register_forward_hook(lambda mod, inp, out, n=n: rec.__setitem__(n, out.detach().float().clone()) if torch.is_tensor(out) else None))
    with torch.no_grad():
        out = m(x.to(dev))
    rec["OUT"] = out.float()
    for h in hooks: h.remove()
    recs[name] = rec
for k in recs["f32"]:
    if k not in recs["bf16"]: continue
    a, b = recs["f32"][k], recs["bf16"][k]
    if a.shape != b.shape: continue
    print(f"{k:40s} {tuple(a.shape)!s:22s} std {a.std().item():9.4f}  rms err/std {(((a - b) ** 2).mean().sqrt() / a.std()).item():.3e}  max err/max {((a - b).abs().max() / a.abs().max()).item():.3e}")
