"""Where does the bf16 mode's error against the f32 parity mode come from at full module size (cfg3: C=512, H=8, depth 6, 224^2, T=16)?
Stage-by-stage relative error (max |bf16 - f32| / max |f32|) with the same weights and clip: backbone output, patch tokens, residual stream
after every block, the cls rows before / after the final LayerNorm, the pre_logits feature, KTD outputs.  The f32 mode agrees with the CPU
oracle to ~2e-6 (tests/test_gpu_model.py::test_cfg2_full_size_forward_f32_vs_oracle), so this is the error against the reference.
usage: diag_bf16_error.py [scale of cls/pos/temp embeddings, default 1 = the reference's init (std 0.02)]"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MAED_SYNTHETIC_SMPL_OK", "1")
import maed_amd
from oracle import maed_ref as R
emb_scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dev = torch.device("cuda", 0)
C, H, depth, img, T, hidden = 512, 8, 6, 224, 16, 1024
params = R.make_params(embed_dim=C, depth=depth, hidden_dim=hidden, n_tokens=(img // 16) ** 2 + 1, seed=7)
for k in ("encoder.cls_token", "encoder.pos_embed", "encoder.temp_embed"):
    params[k] = params[k] * emb_scale
clip = torch.randn(1, T, 3, img, img, generator=torch.Generator().manual_seed(21))
stages = {}
for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
    m = maed_amd.MAED(num_blocks=depth, num_heads=H, embed_dim=C, hidden_dim=hidden, img_size=img, compute_dtype=dt)
    m.load_state_dict(params, strict=False)
    m = m.to(dev).eval()
    enc = m.encoder
    rec = {}
    with torch.no_grad():
        x = clip.reshape(-1, 3, img, img).to(dev)
        rec["backbone out"] = enc.patch_embed.backbone(x).float()
        patch = enc.patch_embed(x)
        rec["patch tokens"] = patch.float()
        from maed_amd import ops
        tok = ops.EmbedAddFn.apply(patch, enc.cls_token, enc.pos_embed, enc.temp_embed, T)
        rec["tokens + embeddings"] = tok.clone()
        for i, blk in enumerate(enc.blocks):
            tok = blk(tok, T)
            rec[f"residual stream after block {i}"] = tok.clone()
        cls = tok[:, 0]
        rec["cls rows (input of the final LayerNorm)"] = cls.clone()
        rec["cls rows: deviation from their channel mean"] = cls - cls.mean(-1, keepdim=True)
        y = F.layer_norm(cls, (C,), enc.norm.weight, enc.norm.bias, enc.norm.eps)
        rec["after final LayerNorm"] = y
        rec["pre_logits feature"] = m.extract_feature(clip.to(dev)).float().reshape(T, C)
        out = m(clip.to(dev))
        for k in ("theta", "kp_3d", "rotmat"):
            rec["output " + k] = out[k].float()
        rec["theta[cam]"] = out["theta"][..., :3].float(); rec["theta[pose]"] = out["theta"][..., 3:75].float(); rec["theta[shape]"] = out["theta"][..., 75:].float()
    stages[name] = rec
print(f"embedding scale x{emb_scale}")
for k in stages["f32"]:
    a, b = stages["f32"][k], stages["bf16"][k]
    print(f"{k:50s} max|f32| {a.abs().max().item():10.4f}  std {a.std().item():10.4f}   rel err {((a - b).abs().max() / a.abs().max()).item():.3e}   rms err / std {(((a - b) ** 2).mean().sqrt() / a.std()).item():.3e}")
