"""GPU parity of the decoder-tail training kernels (maed_amd/csrc/tail_bwd.hip through maed_amd/tail.py) and of the
fused loss (loss.hip through maed_amd/loss.py), on the real library:
  * tail forward + backward against the ATen composition of the same graph at BASELINE's decoder size
    (feat 512, hidden 1024, F = 128 frames) -- that composition is tied to the fp64 oracle in test_gpu_model.py;
  * fused loss values and gradients against the fixtures produced by the reference's own lib/core/loss.py.
"""
import os

import numpy as np
import pytest
import torch

from _util import DEV, report

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def make_ktd(feat, hidden, seed=0):
    from maed_amd.ktd import KTD
    torch.manual_seed(seed)
    ktd = KTD(feat_dim=feat, hidden_dim=hidden).eval()        # Dropout off: both paths see the same graph
    for r in ktd._regressors():
        torch.nn.init.normal_(r.weight, std=0.02)
        torch.nn.init.normal_(r.bias, std=0.3)
    with torch.no_grad():
        ktd.deccam.bias.copy_(torch.tensor([0.9, 0.05, -0.05]))
    return ktd.to(DEV)


@pytest.mark.parametrize("F,feat,hidden,keys", [(128, 512, 1024, ("theta", "kp_2d", "kp_3d")), (5, 64, 32, ("theta", "verts", "kp_2d", "kp_3d", "rotmat")),
                                                (1, 64, 32, ("kp_2d",))])
def test_tail_training_path_vs_aten(F, feat, hidden, keys):
    ktd = make_ktd(feat, hidden)
    params = list(ktd.parameters())
    g = torch.Generator().manual_seed(2)
    x = torch.randn(F, feat, generator=g).to(DEV).requires_grad_(True)
    assert ktd._use_hip_train(x, None)
    out_ref = ktd.get_output(*ktd._head_torch(x), None, hip=False)
    cot = {k: torch.randn(out_ref[k].shape, generator=g).to(DEV) for k in keys}
    gref = torch.autograd.grad(sum((out_ref[k] * cot[k]).sum() for k in cot), [x] + params, allow_unused=True)
    for p in params:
        p.grad = None
    out = ktd(x, seqlen=1)
    sum((out[k] * cot[k]).sum() for k in cot).backward()
    for k in out_ref:
        report(f"tail fwd {k} (F={F})", out[k].detach(), out_ref[k].detach(), rtol=1e-4, atol=2e-5)
    worst, worst_name = 0.0, ""
    for (n, p), b in zip([("x", x)] + list(ktd.named_parameters()), gref):
        if b is None:
            continue
        assert p.grad is not None, n
        e = (p.grad - b).abs().max().item() / (b.abs().max().item() + 1e-12)
        if e > worst:
            worst, worst_name = e, n
    report(f"tail bwd: worst rel-to-max gradient error vs ATen autograd (F={F}, worst {worst_name})", torch.tensor([worst]), torch.zeros(1), rtol=0, atol=5e-4)


@pytest.mark.parametrize("F,feat,hidden", [(16, 512, 1024), (3, 64, 32)])
def test_tail_training_path_vs_fp64_oracle(F, feat, hidden):
    """the same forward + backward against the ORACLE itself (oracle/maed_ref.py ktd_head + ktd_get_output: rot6d -> SMPL LBS -> 49 joints -> projection,
    evaluated in fp64 with autograd on the host) instead of the package's own ATen composition: outputs and the gradients w.r.t. the input features and every KTD
    parameter, cotangents on theta / kp_2d / kp_3d / verts / rotmat"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import maed_ref as R
    ktd = make_ktd(feat, hidden)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(F, feat, generator=g)
    pd = {"d." + k: v.detach().cpu().double().requires_grad_(True) for k, v in ktd.state_dict().items() if not k.startswith("smpl.")}
    xd = x.double().requires_grad_(True)
    ref = R.ktd_get_output(*R.ktd_head(xd, pd, "d."), R.make_synthetic_smpl(0, dtype=torch.float64))
    keys = ("theta", "kp_2d", "kp_3d", "verts", "rotmat")
    cot = {k: torch.randn(ref[k].shape, generator=g) for k in keys}
    sum((ref[k] * cot[k].double()).sum() for k in keys).backward()
    xg = x.to(DEV).requires_grad_(True)
    for p_ in ktd.parameters():
        p_.grad = None
    assert ktd._use_hip_train(xg, None)
    out = ktd(xg, seqlen=1)
    sum((out[k] * cot[k].to(DEV)).sum() for k in keys).backward()
    for k in keys:
        report(f"tail vs oracle fwd {k} (F={F}, feat {feat})", out[k].detach().reshape(ref[k].shape), ref[k].detach(), rtol=1e-4, atol=1e-4 * ref[k].abs().max().item())
    worst, worst_name = 0.0, ""
    for n, p_ in [("x", xg)] + list(ktd.named_parameters()):
        b = xd.grad if n == "x" else pd["d." + n].grad
        if b is None:
            continue
        assert p_.grad is not None, n
        e = (p_.grad.detach().cpu().double() - b).abs().max().item() / (b.abs().max().item() + 1e-30)
        if e > worst:
            worst, worst_name = e, n
    report(f"tail vs oracle bwd: worst rel-to-max gradient error over x and all KTD parameters (F={F}, worst {worst_name})", torch.tensor([worst]), torch.zeros(1), rtol=0, atol=1e-3)


def test_tail_joint_gather_backward_is_scatter_add():
    """joint_map maps several of the 49 outputs onto the same source joint (smpl.py:16-53): the backward must SUM them."""
    from maed_amd import _lib as L, ops
    from maed_amd.smpl import SMPL
    smpl = SMPL().to(DEV)
    F = 3
    g = torch.Generator().manual_seed(5)
    kp3d = torch.randn(F, 49, 3, generator=g).to(DEV)
    cam = (torch.randn(F, 3, generator=g) * 0.1 + torch.tensor([0.9, 0., 0.])).to(DEV)
    d_kp3d = torch.randint(-8, 9, (F, 49, 3), generator=g).float().to(DEV)      # small integers: the sums are exact
    d_j24, d_e21, d_e9, d_cam = [torch.empty(F, n, 3, device=DEV) for n in (24, 21, 9)] + [torch.empty(F, 3, device=DEV)]
    L.check(L.lib().maed_smpl_joints_project_bwd(ops._p(kp3d), ops._p(cam), ops._p(smpl.joint_map), ops._p(d_kp3d), None, None, 0,
                                                  ops._p(d_j24), ops._p(d_e21), ops._p(d_e9), ops._p(d_cam), F, ops._stream()))
    ref = torch.zeros(F, 54, 3, device=DEV).index_add_(1, smpl.joint_map, d_kp3d)
    got = torch.cat([d_j24, d_e21, d_e9], 1)
    assert torch.equal(got, ref), "integer scatter-add through joint_map must be bit-exact"
    assert torch.equal(d_cam, torch.zeros_like(d_cam))
    report("joint_map scatter-add backward (integer cotangents)", got, ref, rtol=0, atol=0)


CASES = ["video_2d3d", "video_3d", "video_novalid", "image", "image_mixed"]


@pytest.mark.parametrize("name", CASES)
def test_fused_loss_matches_reference_golden(name):
    from maed_amd import loss as mloss
    fx = np.load(os.path.join(GOLD, "g11_loss.npz"))
    t = lambda k: torch.from_numpy(fx[k]).to(DEV)
    leaves = {k: t(f"{name}.pred.{k}").requires_grad_(True) for k in ("kp_2d", "kp_3d", "theta")}
    d3 = {k: t(f"{name}.d3.{k}") for k in ("kp_2d", "kp_3d", "theta", "w_smpl")}
    d2 = {"kp_2d": t(f"{name}.d2.kp_2d")} if f"{name}.d2.kp_2d" in fx else None
    if name.startswith("image"):
        total, terms = mloss.Loss().loss_image(leaves, d3)
    else:
        total, terms = mloss.LossVideo()(leaves, d3, d2)
    total.backward()
    assert list(terms.keys()) == list(fx[f"{name}.term_order"])
    report(f"fused loss total [{name}]", total.detach().reshape(1), torch.from_numpy(fx[f"{name}.total"]).reshape(1), rtol=1e-4, atol=1e-6)
    for k, v in terms.items():
        report(f"fused loss {k} [{name}]", v.detach().reshape(1), torch.from_numpy(fx[f"{name}.term.{k}"]).reshape(1), rtol=1e-4, atol=1e-6)
    for k, v in leaves.items():
        ref = torch.from_numpy(fx[f"{name}.grad.{k}"])
        report(f"fused loss d/d{k} [{name}]", v.grad, ref, rtol=1e-4, atol=1e-5 * max(1.0, ref.abs().max().item()))


def test_fused_loss_full_size_properties():
    """BASELINE size (8 clips x 16 frames): the fused kernels against the ATen mirror of the same module (itself pinned to
    the reference on CPU), and linearity of the gradient in the upstream scalar."""
    from maed_amd import loss as mloss
    g = torch.Generator().manual_seed(9)
    r = lambda *s: torch.randn(*s, generator=g)
    N, T = 8, 16
    preds = dict(kp_2d=r(N, T, 49, 2) * 0.5, kp_3d=r(N, T, 49, 3) * 0.4, theta=torch.cat([r(N, T, 3) * 0.1 + 0.9, r(N, T, 72) * 0.4, r(N, T, 10)], -1))
    d3 = dict(kp_2d=torch.cat([r(N, T, 49, 2), torch.rand(N, T, 49, 1, generator=g)], -1), kp_3d=torch.cat([r(N, T, 49, 3), torch.ones(N, T, 49, 1)], -1),
              theta=r(N, T, 85) * 0.3, w_smpl=(torch.rand(N, T, generator=g) > 0.3).float())
    lv = mloss.LossVideo()
    cpu_leaves = {k: v.clone().requires_grad_(True) for k, v in preds.items()}
    tot_c, terms_c = lv(cpu_leaves, d3, None)
    (3.0 * tot_c).backward()
    gpu_leaves = {k: v.to(DEV).requires_grad_(True) for k, v in preds.items()}
    tot_g, terms_g = lv(gpu_leaves, {k: v.to(DEV) for k, v in d3.items()}, None)
    (3.0 * tot_g).backward()
    report("fused loss total, 8x16 frames vs ATen mirror", tot_g.detach().reshape(1), tot_c.detach().reshape(1), rtol=1e-4, atol=1e-6)
    for k in terms_c:
        report(f"fused loss {k}, 8x16", terms_g[k].detach().reshape(1), terms_c[k].detach().reshape(1), rtol=1e-4, atol=1e-6)
    for k in preds:
        ref = cpu_leaves[k].grad
        report(f"fused loss 3*d/d{k}, 8x16", gpu_leaves[k].grad, ref, rtol=1e-4, atol=1e-5 * ref.abs().max().item())
