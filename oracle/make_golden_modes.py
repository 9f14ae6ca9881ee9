"""Generate tests/golden/g12_iterative.npz and g13_st_modes.npz by running the REFERENCE itself (shim-imported
from /root/reference) -- SURVEY.md 8(f) rank 3: decoder='iterative' and the Attention modes other than 'parallel'.
Runs only in the build container:

    cd /root/repo && python -m oracle.make_golden_modes

g13 reuses the state_dicts already committed in g1_attention / g2_block / g4_vit_tiny (the non-parallel modes have
the same parameters minus attn.ts_attn, and minus temp_embed for 'vanilla'/'temporal'), so it stores only the
reference's outputs and gradients.  Gradients are taken in eval mode against fixed cotangents stored in the fixture.
"""
import os
import sys
from functools import partial

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import maed_ref, ref_shims  # noqa: E402
from oracle.make_golden import OUT, randomize, save, sd_np  # noqa: E402

MODES = ("series", "vanilla", "temporal", "coupling")
ROW_STEP = 4   # weight-matrix gradients are stored as rows [::ROW_STEP]


def load_sd(name, prefix="sd."):
    z = np.load(os.path.join(OUT, name + ".npz"))
    return z, {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def grads_np(module, prefix, row_step=1):
    """parameter gradients; 2-D ones keep every row_step-th row (fixture size)"""
    return {prefix + k: (p.grad[::row_step] if p.dim() == 2 else p.grad).detach().numpy()
            for k, p in module.named_parameters() if p.grad is not None}


def main():
    sp = ref_shims.install(smpl_seed=0)
    g = torch.Generator().manual_seed(77)

    # ---- g12: iterative regressor ------------------------------------------------------------------
    mean = dict(pose=(maed_ref.synthetic_mean_params()["pose"] + 0.1 * torch.randn(144, generator=g)).numpy(),
                shape=(0.1 * torch.randn(10, generator=g)).numpy().astype(np.float32),
                cam=np.array([0.9, 0.05, -0.02], dtype=np.float32))
    shim_load = np.load

    def load_with_mean(path, *a, **k):
        if isinstance(path, str) and path.endswith("smpl_mean_params.npz"):
            return mean
        return shim_load(path, *a, **k)

    np.load = load_with_mean
    import lib.models.spin as spin
    import lib.models.vision_transformer as vt
    import lib.models.resnetv2 as rn

    reg = spin.Regressor(feat_dim=128, hidden_dim=64).eval()
    randomize(reg, 21)
    with torch.no_grad():
        for m in (reg.decpose, reg.decshape, reg.deccam):
            m.weight.mul_(0.2)
            m.bias.mul_(0.1)
    xf = torch.randn(6, 128, generator=g).requires_grad_(True)
    pose, shape, cam = reg.iterative_regress(xf)
    o49 = reg(xf, seqlen=3)
    with torch.no_grad():
        o17 = reg(xf, seqlen=3, J_regressor=sp["J_regressor_h36m"])
    cot = {k: torch.randn(o49[k].shape, generator=g) for k in ("theta", "kp_2d", "kp_3d")}
    sum((o49[k] * cot[k]).sum() for k in cot).backward()
    sd = {k: v for k, v in sd_np(reg, "sd.").items() if ".smpl." not in k}
    save("g12_iterative", x=xf.detach(), pose6d=pose.detach(), shape=shape.detach(), cam=cam.detach(),
         theta=o49["theta"].detach(), kp_2d=o49["kp_2d"].detach(), kp_3d=o49["kp_3d"].detach(), rotmat=o49["rotmat"].detach(),
         verts_sub=o49["verts"].detach()[:, ::53], kp_3d_h36m=o17["kp_3d"], smpl_seed=0,
         mean_pose=mean["pose"], mean_shape=mean["shape"], mean_cam=mean["cam"],
         cot_theta=cot["theta"], cot_kp_2d=cot["kp_2d"], cot_kp_3d=cot["kp_3d"], gx=xf.grad,
         **{k: v for k, v in grads_np(reg, "grad.").items() if ".smpl." not in k}, **sd)

    # ---- g13: Attention / Block / tiny ViT in the other st_modes -------------------------------------
    LN = partial(nn.LayerNorm, eps=1e-6)
    z1, sd1 = load_sd("g1_attention")
    z2, sd2 = load_sd("g2_block")
    z4, sd4 = load_sd("g4_vit_tiny")
    x = torch.from_numpy(z1["x"])
    T, H = int(z1["seqlen"]), int(z1["heads"])
    img = torch.from_numpy(z4["img"])
    fx = dict(row_step=ROW_STEP, cot_tok=torch.randn(x.shape, generator=g).numpy(), cot_feat=torch.randn(img.shape[0], 128, generator=g).numpy())
    for mode in MODES:
        att = vt.Attention(128, num_heads=H, qkv_bias=True, st_mode=mode).eval()
        att.load_state_dict({k: v for k, v in sd1.items() if not k.startswith("ts_attn")})
        xa = x.clone().requires_grad_(True)
        out = att(xa, T)
        (out * torch.from_numpy(fx["cot_tok"])[:, :out.shape[1]]).sum().backward()
        fx.update({f"{mode}.att.out": out.detach().numpy(), f"{mode}.att.dx": xa.grad.numpy()})
        fx.update(grads_np(att, f"{mode}.att.grad.", ROW_STEP))

        blk = vt.Block(128, H, mlp_ratio=4, qkv_bias=True, norm_layer=LN, st_mode=mode).eval()
        blk.load_state_dict({k: v for k, v in sd2.items() if "ts_attn" not in k})
        xb = x.clone().requires_grad_(True)
        out = blk(xb, T)
        (out * torch.from_numpy(fx["cot_tok"])).sum().backward()
        fx.update({f"{mode}.blk.out": out.detach().numpy(), f"{mode}.blk.dx": xb.grad.numpy()})
        fx.update(grads_np(blk, f"{mode}.blk.grad.", ROW_STEP))

        bb = rn.ResNetV2(layers=(1, 1, 1), channels=(128, 256, 512), num_classes=0, global_pool="", in_chans=3, preact=False,
                         stem_type="same")
        vit = vt.VisionTransformer(img_size=32, patch_size=16, embed_dim=128, depth=2, num_heads=2, hybrid_backbone=bb, mlp_ratio=4,
                                   qkv_bias=True, representation_size=128, norm_layer=LN, st_mode=mode, num_classes=-1).eval()
        keep = {k: v for k, v in sd4.items() if "ts_attn" not in k and (k != "temp_embed" or hasattr(vit, "temp_embed"))}
        missing, unexpected = vit.load_state_dict(keep, strict=False)
        assert not missing and not unexpected, (mode, missing, unexpected)
        feat = vit(img, seqlen=int(z4["seqlen"]))
        (feat * torch.from_numpy(fx["cot_feat"])).sum().backward()
        fx[f"{mode}.vit.out"] = feat.detach().numpy()
        fx[f"{mode}.vit.has_temp_embed"] = hasattr(vit, "temp_embed")
        # a few gradients of the whole encoder: embeddings, first block's attention, last norm
        for k, p in vit.named_parameters():
            if k in ("cls_token", "pos_embed", "temp_embed", "norm.weight", "norm.bias", "blocks.0.attn.qkv.bias",
                     "blocks.0.attn.proj.weight", "blocks.1.mlp.fc1.bias", "patch_embed.proj.bias"):
                fx[f"{mode}.vit.grad.{k}"] = p.grad.numpy()
    save("g13_st_modes", **fx)


if __name__ == "__main__":
    main()
