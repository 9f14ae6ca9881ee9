"""Generate tests/golden/*.npz by running the REFERENCE itself (shim-imported from
/root/reference) on seeded inputs.  Runs only in the build container:

    cd /root/repo && python -m oracle.make_golden

Each fixture stores inputs, the module's state_dict (tiny modules) and the reference's outputs,
so neither the tests nor the GPU box ever need reference code.  For the full-size cfg1 fixture
the (288 MB) weights are not stored: they are regenerated deterministically with
oracle.maed_ref.make_params(seed) both here (loaded INTO the reference model) and in the test.
"""
import os
import sys
from functools import partial

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import maed_ref, ref_shims  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def sd_np(module, prefix=""):
    return {prefix + k: v.detach().numpy() for k, v in module.state_dict().items()}


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def randomize(module, seed):
    """make every parameter non-trivial (LN weights != 1, biases != 0) so parity is meaningful"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.dim() == 1:
                base = 1.0 if n.endswith("weight") else 0.0
                p.copy_(base + 0.2 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / max(1, p[0].numel()) ** 0.5))


@torch.no_grad()
def main():
    sp = ref_shims.install(smpl_seed=0)
    import lib.models.vision_transformer as vt
    import lib.models.resnetv2 as rn
    import lib.models.ktd as ktd
    import lib.models.smpl as smpl_mod
    import lib.models.spin as spin
    import lib.utils.geometry as geo
    from lib.models.maed import MAED

    LN = partial(nn.LayerNorm, eps=1e-6)
    g = torch.Generator().manual_seed(1)

    # g1: Attention (parallel) --------------------------------------------------------------
    att = vt.Attention(128, num_heads=2, qkv_bias=True, st_mode="parallel").eval()
    randomize(att, 11)
    x = torch.randn(6, 5, 128, generator=g)
    out = att(x, 3)
    qkv = att.qkv(x).reshape(6, 5, 3, 2, 64).permute(2, 0, 3, 1, 4)
    x_t = att.forward_temporal(qkv[0], qkv[1], qkv[2], seqlen=3)
    x_s = att.forward_spatial(qkv[0], qkv[1], qkv[2])
    save("g1_attention", x=x, seqlen=3, heads=2, out=out, x_s=x_s, x_t=x_t, **sd_np(att, "sd."))

    # g2: Block -----------------------------------------------------------------------------
    blk = vt.Block(128, 2, mlp_ratio=4, qkv_bias=True, norm_layer=LN, st_mode="parallel").eval()
    randomize(blk, 12)
    save("g2_block", x=x, seqlen=3, heads=2, out=blk(x, 3), **sd_np(blk, "sd."))

    # g3: Mlp + LayerNorm -------------------------------------------------------------------
    m = vt.Mlp(128, 512).eval()
    randomize(m, 13)
    ln = LN(128)
    randomize(ln, 14)
    save("g3_mlp_ln", x=x, mlp_out=m(x), ln_out=ln(x * 3 + 0.5),
         **sd_np(m, "mlp."), **sd_np(ln, "ln."))

    # g4: tiny VisionTransformer with hybrid ResNetV2(1,1,1), 32x32 input -> P=5 -------------
    bb = rn.ResNetV2(layers=(1, 1, 1), channels=(128, 256, 512), num_classes=0, global_pool="", in_chans=3, preact=False, stem_type="same")
    vit = vt.VisionTransformer(img_size=32, patch_size=16, embed_dim=128, depth=2, num_heads=2, hybrid_backbone=bb,
                               mlp_ratio=4, qkv_bias=True, representation_size=128, norm_layer=LN,
                               st_mode="parallel", num_classes=-1).eval()
    randomize(vit, 15)
    img = torch.randn(4, 3, 32, 32, generator=g)
    feat_map = vit.patch_embed.backbone(img) if False else vit.patch_embed.backbone.forward_features(img)
    tokens = vit.patch_embed(img)
    save("g4_vit_tiny", img=img, seqlen=2, heads=2, depth=2, layers=np.array([1, 1, 1]),
         backbone_out=feat_map, tokens=tokens, out=vit(img, seqlen=2), **sd_np(vit, "sd."))

    # g5: backbone pieces on odd/even sizes ---------------------------------------------------
    fx = {}
    for tag, size in (("odd", 17), ("even", 16)):
        xi = torch.randn(2, 32, size, size, generator=g)
        c3 = rn.StdConv2dSame(32, 64, 3, stride=2)
        c7 = rn.StdConv2dSame(32, 64, 7, stride=2)
        c1 = rn.StdConv2dSame(32, 64, 1, stride=1)
        for c in (c3, c7, c1):
            randomize(c, 16)
        gn = rn.GroupNormAct(64, 32)
        randomize(gn, 17)
        mp = rn.MaxPool2dSame(3, 2)
        bt = rn.Bottleneck(32, 128, stride=2, proj_layer=rn.DownsampleConv,
                           conv_layer=rn.StdConv2dSame, norm_layer=partial(rn.GroupNormAct, num_groups=32)).eval()
        randomize(bt, 18)
        y3 = c3(xi)
        fx.update({f"{tag}.x": xi, f"{tag}.conv3s2": y3, f"{tag}.conv7s2": c7(xi), f"{tag}.conv1": c1(xi),
                   f"{tag}.gn_relu": gn(y3), f"{tag}.maxpool": mp(xi), f"{tag}.bottleneck": bt(xi)})
        fx.update({f"{tag}.w3": c3.weight.numpy(), f"{tag}.w7": c7.weight.numpy(), f"{tag}.w1": c1.weight.numpy(),
                   f"{tag}.gn.weight": gn.weight.numpy(), f"{tag}.gn.bias": gn.bias.numpy()})
        fx.update(sd_np(bt, f"{tag}.bt."))
    save("g5_backbone_pieces", **fx)

    # g6: KTD head + get_output (stub SMPL = oracle LBS on synthetic params, seed 0) ---------
    dec = ktd.KTD(feat_dim=128, hidden_dim=64).eval()
    randomize(dec, 19)
    with torch.no_grad():
        dec.deccam.bias.copy_(torch.tensor([0.9, 0.05, -0.1]))
    xf = torch.randn(6, 128, generator=g)
    captured = {}
    orig = dec.get_output

    def capture(pose, shape, cam, J):
        captured.update(pose=pose, shape=shape, cam=cam)
        return orig(pose, shape, cam, J)

    dec.get_output = capture
    o49 = dec(xf, seqlen=3)
    o17 = dec(xf, seqlen=3, J_regressor=sp["J_regressor_h36m"])
    sd = {k: v for k, v in sd_np(dec, "sd.").items() if ".smpl." not in k}
    save("g6_ktd", x=xf, pose6d=captured["pose"], shape=captured["shape"], cam=captured["cam"],
         theta=o49["theta"], kp_2d=o49["kp_2d"], kp_3d=o49["kp_3d"], rotmat=o49["rotmat"],
         verts_sub=o49["verts"][:, ::53], kp_2d_h36m=o17["kp_2d"], kp_3d_h36m=o17["kp_3d"],
         joint_map=dec.smpl.joint_map.numpy(), smpl_seed=0, **sd)

    # g7: geometry -----------------------------------------------------------------------------
    r6 = torch.randn(64, 6, generator=g)
    R = geo.rot6d_to_rotmat(r6)
    # craft rotations that hit each quaternion branch: rotations by ~pi about x, y, z and small ones
    def axis_rot(axis, ang):
        a = torch.zeros(3)
        a[axis] = ang
        return geo.batch_rodrigues(a[None]).view(3, 3)

    special = torch.stack([axis_rot(0, 3.0), axis_rot(1, 3.0), axis_rot(2, 3.0), axis_rot(0, 1e-4),
                           axis_rot(1, 0.5), axis_rot(2, -2.5), torch.eye(3), axis_rot(0, 3.14159)])
    Rall = torch.cat([R, special])
    aa = geo.rotation_matrix_to_angle_axis(Rall)
    joints = torch.randn(5, 49, 3, generator=g)
    cam = torch.tensor([[0.9, 0.1, -0.2], [1.1, 0.0, 0.0], [0.7, -0.3, 0.2], [1.0, 0.2, 0.1], [0.8, 0.0, -0.1]])
    axang = torch.randn(16, 3, generator=g)
    save("g7_geometry", rot6d=r6, rotmat=R, rotmat_all=Rall, angle_axis=aa, joints=joints, cam=cam,
         kp_2d=spin.projection(joints, cam), axisang=axang, rodrigues=geo.batch_rodrigues(axang))

    # g9: joint map (integers) -----------------------------------------------------------------
    jm = [smpl_mod.JOINT_MAP[n] for n in smpl_mod.JOINT_NAMES]
    save("g9_joint_map", joint_map=np.array(jm, dtype=np.int64), ancestor_len=np.array([len(a) for a in ktd.ANCESTOR_INDEX]),
         ancestors_flat=np.array(sum(ktd.ANCESTOR_INDEX, []), dtype=np.int64))

    # g10: full-size cfg1 (N=2,T=8, 224^2, C=768, H=12, 6 blocks) through the reference factory ---
    torch.manual_seed(0)
    model = MAED(encoder="ste", num_blocks=6, num_heads=12, st_mode="parallel", decoder="ktd", hidden_dim=1024).eval()
    params = maed_ref.make_params(embed_dim=768, depth=6, hidden_dim=1024, seed=7)
    missing, unexpected = model.load_state_dict(params, strict=False)
    assert not unexpected, unexpected
    assert all(".smpl." in k for k in missing), missing
    clip = torch.randn(2, 8, 3, 224, 224, generator=torch.Generator().manual_seed(1234))
    feat = model.extract_feature(clip)  # note: extract_feature omits seqlen (maed.py:47) -> use forward for outputs
    out = model(clip)
    out17 = model(clip, J_regressor=sp["J_regressor_h36m"])
    save("g10_cfg1_full", param_seed=7, clip_seed=1234, smpl_seed=0, n_params=sum(p.numel() for p in model.parameters()),
         theta=out["theta"], kp_2d=out["kp_2d"], kp_3d=out["kp_3d"], rotmat=out["rotmat"],
         verts_sub=out["verts"][:, :, ::53], kp_3d_h36m=out17["kp_3d"],
         state_dict_keys=np.array(sorted(model.state_dict().keys())))
    del feat


if __name__ == "__main__":
    main()
