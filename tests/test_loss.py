"""lib/core/loss.py parity: oracle and the host-side mirror against fixtures produced by the
reference's own classes (oracle/make_golden_loss.py), and the fused HIP loss kernels on the host simulator."""
import os

import numpy as np
import pytest
import torch

from maed_amd import loss as mloss
from oracle import loss_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["video_2d3d", "video_3d", "video_accl", "video_novalid", "image", "image_mixed"]   # image_mixed: w_smpl has zeros (not masked, loss.py:77)
ACCL_KW = dict(e_loss_weight=5., e_3d_loss_weight=7., e_pose_loss_weight=2., e_shape_loss_weight=0.5, e_smpl_norm_loss=0.25, e_smpl_accl_loss=3.)


def load_case(fx, name, dtype=torch.float32):
    t = lambda k: torch.from_numpy(fx[k]).to(dtype)
    preds = {k: t(f"{name}.pred.{k}") for k in ("kp_2d", "kp_3d", "theta")}
    d3 = {k: t(f"{name}.d3.{k}") for k in ("kp_2d", "kp_3d", "theta", "w_smpl")}
    d2 = {"kp_2d": t(f"{name}.d2.kp_2d")} if f"{name}.d2.kp_2d" in fx else None
    return preds, d3, d2


def check(fx, name, total, terms, grads, rtol, atol):
    assert list(terms.keys()) == list(fx[f"{name}.term_order"])
    np.testing.assert_allclose(float(total.detach()), fx[f"{name}.total"], rtol=rtol, atol=atol)
    for k, v in terms.items():
        np.testing.assert_allclose(float(v.detach()), fx[f"{name}.term.{k}"], rtol=rtol, atol=atol, err_msg=k)
    for k, g in grads.items():
        ref = fx[f"{name}.grad.{k}"]
        np.testing.assert_allclose(g.numpy(), ref, rtol=rtol, atol=atol * max(1.0, np.abs(ref).max()), err_msg=k)


def test_rodrigues_golden():
    fx = np.load(os.path.join(GOLD, "g8_rodrigues.npz"))
    aa = torch.from_numpy(fx["axis_angle"])
    np.testing.assert_allclose(loss_ref.batch_rodrigues(aa).numpy(), fx["rotmat"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(mloss.batch_rodrigues(aa).numpy(), fx["rotmat"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(loss_ref.quat2mat(torch.from_numpy(fx["quat"])).numpy(), fx["quat_rotmat"], rtol=0, atol=2e-6)
    assert np.isfinite(fx["rotmat"]).all()


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(name):
    fx = np.load(os.path.join(GOLD, "g11_loss.npz"))
    preds, d3, d2 = load_case(fx, name, torch.float64)
    leaves = {k: v.clone().requires_grad_(True) for k, v in preds.items()}
    if name.startswith("image"):
        total, terms = loss_ref.loss_image(leaves, d3, w3d=30.)       # Loss() hands its e_3d_loss_weight=30 to LossImage
    elif name == "video_accl":
        total, terms = loss_ref.loss_video(leaves, d3, d2, 5., 7., 2., 0.5, 0.25, 3.)
    else:
        total, terms = loss_ref.loss_video(leaves, d3, d2)
    total.backward()
    check(fx, name, total, terms, {k: v.grad for k, v in leaves.items()}, rtol=2e-5, atol=2e-6)


def _module_for(name):
    if name.startswith("image"):
        return mloss.Loss(device="cpu").loss_image
    return mloss.LossVideo(device="cpu", **(ACCL_KW if name == "video_accl" else {}))


@pytest.mark.parametrize("name", CASES)
def test_host_mirror_matches_reference(name):
    fx = np.load(os.path.join(GOLD, "g11_loss.npz"))
    preds, d3, d2 = load_case(fx, name)
    leaves = {k: v.clone().requires_grad_(True) for k, v in preds.items()}
    mod = _module_for(name)
    total, terms = mod(leaves, d3) if name.startswith("image") else mod(leaves, d3, d2)
    total.backward()
    check(fx, name, total, terms, {k: v.grad for k, v in leaves.items()}, rtol=2e-5, atol=2e-6)


def test_front_end_and_merge():
    fx = np.load(os.path.join(GOLD, "g11_loss.npz"))
    front = mloss.Loss(device="cpu")
    pv, d3v, d2v = load_case(fx, "merge")
    pi, d3i, _ = load_case(fx, "image")
    lv, dv = front(pv, target_3d=d3v, target_2d=d2v)
    li, di = front(pi, target_img=d3i)
    lm, dm = front.merge_loss(lv, dv, li, di, vid_w=0.7, img_w=1.3)
    np.testing.assert_allclose(float(lm), fx["merge.total"], rtol=2e-5)
    for k, v in dm.items():
        np.testing.assert_allclose(float(v), fx["merge.term." + k], rtol=2e-5, atol=1e-7, err_msg=k)
    assert front(pv) == (0, {})


@pytest.mark.parametrize("name", ["video_2d3d", "video_3d", "video_novalid", "video_accl", "image", "image_mixed"])
def test_fused_kernels_on_host_simulator(name, monkeypatch):
    """maed_loss_fwd_bwd (maed_amd/csrc/loss.hip) compiled for x86 against tests/hostsim: values AND gradients vs the
    reference's.  The fused path is selected by .is_cuda in the module, so call its back end directly."""
    from _hostsim import patched
    fx = np.load(os.path.join(GOLD, "g11_loss.npz"))
    preds, d3, d2 = load_case(fx, name)
    leaves = {k: v.clone().requires_grad_(True) for k, v in preds.items()}
    with patched():
        if name.startswith("image"):
            total, terms = mloss.Loss(device="cpu").loss_image(leaves, d3)       # on_library_device: the simulator counts as the device
        else:      # (video_accl: the acceleration term's own kernel, maed_loss_accl_fwd_bwd, on top of the fused five)
            total, terms = mloss.LossVideo(device="cpu", **(ACCL_KW if name == "video_accl" else {}))(leaves, d3, d2)
        (total * 1.0).backward()
    check(fx, name, total, terms, {k: v.grad for k, v in leaves.items()}, rtol=3e-5, atol=3e-6)
