"""Is the bf16 error growth of the random-initialised hybrid R50 a property of bf16 arithmetic on this network, or of our kernels?
Same weights and input through (a) the oracle's functional backbone in fp32 on the GPU (ATen), (b) the SAME functional code with every
tensor in bf16 (pure ATen / MIOpen, no libmaed_hip kernel involved), (c) maed_amd.ResNetV2 in bf16.  rms error / std of the output."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maed_amd.resnetv2 import ResNetV2
from oracle import maed_ref as R
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 4
img = 224
dev = torch.device("cuda", 0)
params = R.make_params(embed_dim=512, depth=1, hidden_dim=64, n_tokens=(img // 16) ** 2 + 1, seed=7)
pre = "encoder.patch_embed.backbone."
x = torch.randn(nf, 3, img, img, generator=torch.Generator().manual_seed(21)).to(dev)
with torch.no_grad():
    p32 = {k: v.to(dev) for k, v in params.items() if k.startswith(pre)}
    ref = R.resnetv2_features(x, p32, pre)
    # (b) pure ATen in bf16: weights standardised in fp32 then everything bf16 is what a mixed-precision user would run; here the crudest form
    p16 = {k: v.bfloat16() for k, v in p32.items()}
    aten16 = R.resnetv2_features(x.bfloat16(), p16, pre).float()
    # (b') autocast: convolutions in bf16, normalisations in fp32 (torch.autocast's policy)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        auto16 = R.resnetv2_features(x, p32, pre).float()
    m = ResNetV2(layers=(3, 4, 9), compute_dtype=torch.bfloat16)
    m.load_state_dict({k[len(pre):]: v for k, v in params.items() if k.startswith(pre)})
    mine = m.to(dev).eval()(x).float()
    m32 = ResNetV2(layers=(3, 4, 9), compute_dtype=torch.float32)
    m32.load_state_dict({k[len(pre):]: v for k, v in params.items() if k.startswith(pre)})
    mine32 = m32.to(dev).eval()(x).float()
e = lambda a: (((a - ref) ** 2).mean().sqrt() / ref.std()).item()
print(f"rms error / std of the backbone output vs fp32 ATen:  maed_amd f32 {e(mine32):.3e} | pure-ATen bf16 {e(aten16):.3e} | ATen autocast(bf16) {e(auto16):.3e} | maed_amd bf16 {e(mine):.3e}")
