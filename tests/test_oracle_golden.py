"""Pin the CPU oracle (oracle/maed_ref.py) to the reference: every fixture under tests/golden/
was produced by running the reference's own modules (oracle/make_golden.py).  fp32, CPU."""
import numpy as np
import pytest
import torch

from oracle import maed_ref as R

TOL = dict(rtol=2e-5, atol=2e-5)


def t(a):
    return torch.from_numpy(np.asarray(a))


def sd(fx, prefix):
    return {k[len(prefix):]: t(fx[k]) for k in fx.files if k.startswith(prefix)}


def close(a, b, **kw):
    kw = {**TOL, **kw}
    np.testing.assert_allclose(a.detach().numpy() if torch.is_tensor(a) else a, np.asarray(b), **kw)


def test_g1_attention(golden):
    fx = golden("g1_attention")
    p = sd(fx, "sd.")
    out, parts = R.attention_parallel(t(fx["x"]), p, "", int(fx["heads"]), int(fx["seqlen"]), return_parts=True)
    close(parts["x_s"], fx["x_s"])
    close(parts["x_t"], fx["x_t"])
    close(out, fx["out"])


def test_g2_block(golden):
    fx = golden("g2_block")
    close(R.block(t(fx["x"]), sd(fx, "sd."), "", int(fx["heads"]), int(fx["seqlen"])), fx["out"])


def test_g3_mlp_ln(golden):
    fx = golden("g3_mlp_ln")
    x = t(fx["x"])
    close(R.mlp(x, sd(fx, "mlp."), ""), fx["mlp_out"])
    ln = sd(fx, "ln.")
    close(R.layer_norm(x * 3 + 0.5, ln["weight"], ln["bias"]), fx["ln_out"])


def test_g4_vit_tiny(golden):
    fx = golden("g4_vit_tiny")
    p = sd(fx, "sd.")
    layers = tuple(int(v) for v in fx["layers"])
    img = t(fx["img"])
    close(R.resnetv2_features(img, p, "patch_embed.backbone.", layers), fx["backbone_out"], rtol=1e-4, atol=1e-4)
    close(R.hybrid_embed(img, p, "patch_embed.", layers), fx["tokens"], rtol=1e-4, atol=1e-4)
    out = R.ste_forward_features(img, p, "", int(fx["depth"]), int(fx["heads"]), int(fx["seqlen"]), layers)
    close(out, fx["out"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("tag", ["odd", "even"])
def test_g5_backbone_pieces(golden, tag):
    fx = golden("g5_backbone_pieces")
    x = t(fx[f"{tag}.x"])
    y3 = R.std_conv_same(x, t(fx[f"{tag}.w3"]), 2)
    close(y3, fx[f"{tag}.conv3s2"], rtol=1e-4, atol=1e-4)
    close(R.std_conv_same(x, t(fx[f"{tag}.w7"]), 2), fx[f"{tag}.conv7s2"], rtol=1e-4, atol=1e-4)
    close(R.std_conv_same(x, t(fx[f"{tag}.w1"]), 1), fx[f"{tag}.conv1"], rtol=1e-4, atol=1e-4)
    close(R.group_norm_act(y3, t(fx[f"{tag}.gn.weight"]), t(fx[f"{tag}.gn.bias"])), fx[f"{tag}.gn_relu"], rtol=1e-4, atol=1e-4)
    close(R.max_pool_same(x), fx[f"{tag}.maxpool"], rtol=0, atol=0)
    close(R.bottleneck(x, sd(fx, f"{tag}.bt."), "", 2), fx[f"{tag}.bottleneck"], rtol=1e-4, atol=1e-4)


def test_g6_ktd(golden):
    fx = golden("g6_ktd")
    p = sd(fx, "sd.")
    sp = R.make_synthetic_smpl(int(fx["smpl_seed"]))
    pose, shape, cam = R.ktd_head(t(fx["x"]), p, "")
    close(pose, fx["pose6d"])
    close(shape, fx["shape"])
    close(cam, fx["cam"])
    o = R.ktd_get_output(t(fx["pose6d"]), t(fx["shape"]), t(fx["cam"]), sp)
    close(o["rotmat"], fx["rotmat"])
    close(o["theta"], fx["theta"])
    close(o["verts"][:, ::53], fx["verts_sub"])
    close(o["kp_3d"], fx["kp_3d"])
    close(o["kp_2d"], fx["kp_2d"], rtol=1e-4, atol=1e-4)
    o17 = R.ktd_get_output(t(fx["pose6d"]), t(fx["shape"]), t(fx["cam"]), sp, sp["J_regressor_h36m"])
    close(o17["kp_3d"], fx["kp_3d_h36m"])
    close(o17["kp_2d"], fx["kp_2d_h36m"], rtol=1e-4, atol=1e-4)
    assert np.array_equal(fx["joint_map"], np.array(R.JOINT_MAP_49))  # integer work: bit-exact


def test_g7_geometry(golden):
    fx = golden("g7_geometry")
    close(R.rot6d_to_rotmat(t(fx["rot6d"])), fx["rotmat"], rtol=1e-6, atol=1e-6)
    close(R.rotmat_to_angle_axis(t(fx["rotmat_all"])), fx["angle_axis"], rtol=1e-5, atol=1e-6)
    close(R.projection(t(fx["joints"]), t(fx["cam"])), fx["kp_2d"], rtol=1e-5, atol=1e-5)


def test_g9_joint_map_and_tree(golden):
    fx = golden("g9_joint_map")
    assert np.array_equal(fx["joint_map"], np.array(R.JOINT_MAP_49, dtype=np.int64))
    assert np.array_equal(fx["ancestor_len"], np.array([len(a) for a in R.ANCESTOR_INDEX]))
    assert np.array_equal(fx["ancestors_flat"], np.array(sum(R.ANCESTOR_INDEX, []), dtype=np.int64))
    assert R.SMPL_PARENTS == [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]


def test_g10_cfg1_full(golden):
    """Whole MAED forward at the reference's true dims (cfg1: 2x8x224^2, C=768, H=12, 6 blocks)."""
    fx = golden("g10_cfg1_full")
    p = R.make_params(embed_dim=768, depth=6, hidden_dim=1024, seed=int(fx["param_seed"]))
    assert sum(v.numel() for v in p.values()) == int(fx["n_params"]) == 72132153
    keys = {k for k in fx["state_dict_keys"].tolist() if ".smpl." not in k}
    assert keys == set(p.keys())
    sp = R.make_synthetic_smpl(int(fx["smpl_seed"]))
    clip = torch.randn(2, 8, 3, 224, 224, generator=torch.Generator().manual_seed(int(fx["clip_seed"])))
    with torch.no_grad():
        o = R.maed_forward(clip, p, sp, depth=6, H=12)
        o17 = R.maed_forward(clip, p, sp, depth=6, H=12, J_regressor=sp["J_regressor_h36m"])
    close(o["theta"], fx["theta"], rtol=1e-3, atol=1e-4)   # north_star tolerance on SMPL params
    close(o["rotmat"], fx["rotmat"], rtol=1e-3, atol=1e-4)
    close(o["kp_3d"], fx["kp_3d"], rtol=1e-3, atol=1e-4)
    close(o["kp_2d"], fx["kp_2d"], rtol=1e-3, atol=1e-3)
    close(o["verts"][:, :, ::53], fx["verts_sub"], rtol=1e-3, atol=1e-4)
    close(o17["kp_3d"], fx["kp_3d_h36m"], rtol=1e-3, atol=1e-4)


# ---- the other st_modes and decoder='iterative' --------------------------------------------------
MODES = ["series", "vanilla", "temporal", "coupling"]


@pytest.mark.parametrize("mode", MODES)
def test_g13_attention_and_block_modes(golden, mode):
    fx, g1, g2 = golden("g13_st_modes"), golden("g1_attention"), golden("g2_block")
    x, H, T = t(g1["x"]), int(g1["heads"]), int(g1["seqlen"])
    close(R.attention_mode(x, sd(g1, "sd."), "", H, T, mode), fx[f"{mode}.att.out"])
    close(R.block(x, sd(g2, "sd."), "", H, T, mode), fx[f"{mode}.blk.out"])


@pytest.mark.parametrize("mode", MODES)
def test_g13_vit_tiny_modes(golden, mode):
    fx, g4 = golden("g13_st_modes"), golden("g4_vit_tiny")
    p = sd(g4, "sd.")
    if not bool(fx[f"{mode}.vit.has_temp_embed"]):
        del p["temp_embed"]                      # the reference has no such parameter in 'vanilla'/'temporal' (:363)
    layers = tuple(int(v) for v in g4["layers"])
    out = R.ste_forward_features(t(g4["img"]), p, "", int(g4["depth"]), int(g4["heads"]), int(g4["seqlen"]), layers, mode)
    close(out, fx[f"{mode}.vit.out"], rtol=1e-4, atol=1e-4)


def test_g12_iterative(golden):
    fx = golden("g12_iterative")
    p = sd(fx, "sd.")
    sp = R.make_synthetic_smpl(int(fx["smpl_seed"]))
    pose, shape, cam = R.iterative_head(t(fx["x"]), p, "", t(fx["mean_pose"])[None], t(fx["mean_shape"])[None], t(fx["mean_cam"])[None])
    close(pose, fx["pose6d"])
    close(shape, fx["shape"])
    close(cam, fx["cam"])
    o = R.ktd_get_output(pose, shape, cam, sp)
    for k in ("theta", "kp_2d", "kp_3d", "rotmat"):
        close(o[k], fx[k], rtol=1e-4, atol=1e-4)
    close(o["verts"][:, ::53], fx["verts_sub"], rtol=1e-4, atol=1e-4)
    o17 = R.ktd_get_output(pose, shape, cam, sp, J_regressor=sp["J_regressor_h36m"])
    close(o17["kp_3d"], fx["kp_3d_h36m"], rtol=1e-4, atol=1e-4)
