"""decoder='iterative' (maed_amd/iterative.py; reference lib/models/spin.py:17-110) against the fixture the reference's own
Regressor produced (tests/golden/g12_iterative.npz, oracle/make_golden_modes.py): the ATen path on CPU tensors, and the
libmaed_hip inference / training paths with the kernels running on the host simulator."""
import numpy as np
import pytest
import torch

from maed_amd import tail
from maed_amd.spin import Regressor

from _hostsim import patched


def t(a):
    return torch.from_numpy(np.asarray(a))


def rel(a, b):
    a, b = torch.as_tensor(a), torch.as_tensor(np.asarray(b))
    return float((a.detach() - b).abs().max() / (b.abs().max() + 1e-12))


def build(fx):
    reg = Regressor(smpl_mean_params=dict(pose=fx["mean_pose"], shape=fx["mean_shape"], cam=fx["mean_cam"]), feat_dim=128, hidden_dim=64)
    sd = {k[3:]: t(fx[k]) for k in fx.files if k.startswith("sd.")}
    own = {k for k in reg.state_dict() if not k.startswith("smpl.")}
    assert own == set(sd), own ^ set(sd)                        # same parameter AND buffer names as the reference
    missing, unexpected = reg.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("smpl.") for k in missing)
    return reg.eval()


def check_grads(reg, x, fx, tol):
    assert rel(x.grad, fx["gx"]) < tol
    for n, p in reg.named_parameters():
        want = fx["grad." + n]
        assert p.grad is not None, n
        assert rel(p.grad, want) < tol, (n, rel(p.grad, want))


def cotangent_loss(out, fx):
    return sum((out[k] * t(fx["cot_" + k])).sum() for k in ("theta", "kp_2d", "kp_3d"))


def test_aten_path_matches_reference(golden):
    fx = golden("g12_iterative")
    reg = build(fx)
    x = t(fx["x"]).clone().requires_grad_(True)
    pose, shape, cam = reg.iterative_regress(x)
    for got, key in ((pose, "pose6d"), (shape, "shape"), (cam, "cam")):
        assert rel(got, fx[key]) < 1e-5, key                    # fc1 split into x-part + parameter-part: rounding only
    out = reg(x, seqlen=3)
    for k in ("theta", "kp_2d", "kp_3d", "rotmat"):
        assert rel(out[k], fx[k]) < 1e-4, k
    cotangent_loss(out, fx).backward()
    check_grads(reg, x, fx, 2e-4)
    from maed_amd.smpl import synthetic_smpl_arrays
    with torch.no_grad():
        o17 = reg(t(fx["x"]), seqlen=3, J_regressor=synthetic_smpl_arrays(int(fx["smpl_seed"]))["J_regressor_h36m"])
    assert rel(o17["kp_3d"], fx["kp_3d_h36m"]) < 1e-4


def test_hip_inference_path_on_simulator(golden):
    fx = golden("g12_iterative")
    reg = build(fx)
    x = t(fx["x"])
    with patched(), torch.no_grad():
        pose, shape, cam = reg._regress_hip(x, *reg._init(x.shape[0], None, None, None), 3)
        out = reg.get_output(pose, shape, cam, None, hip=True)
    for got, key in ((pose, "pose6d"), (shape, "shape"), (cam, "cam")):
        assert rel(got, fx[key]) < 1e-5, key
    for k in ("theta", "kp_2d", "kp_3d", "rotmat"):
        assert rel(out[k], fx[k]) < 1e-4, k
    assert rel(out["verts"][:, ::53], fx["verts_sub"]) < 1e-4


def test_hip_training_tail_on_simulator(golden):
    """training graph on the GPU = ATen GEMMs/Dropout for the 3 rounds + tail.SmplTailFn (HIP forward and backward)"""
    fx = golden("g12_iterative")
    reg = build(fx)
    x = t(fx["x"]).clone().requires_grad_(True)
    with patched():
        pose, shape, cam = reg.iterative_regress(x)
        theta, verts, kp2d, kp3d, rotmat = tail.SmplTailFn.apply(pose, shape, cam, reg.smpl)
        out = dict(theta=theta, kp_2d=kp2d, kp_3d=kp3d)
        for k in out:
            assert rel(out[k], fx[k]) < 1e-4, k
        cotangent_loss(out, fx).backward()
    check_grads(reg, x, fx, 5e-4)


def test_n_iter_argument_is_ignored_by_forward_like_the_reference(golden):
    fx = golden("g12_iterative")
    reg = build(fx)
    x = t(fx["x"])
    with torch.no_grad():
        a = reg(x, seqlen=3, n_iter=1)["theta"]                 # spin.py:83 hard-codes n_iter=3
        one = reg.iterative_regress(x, n_iter=1)[0]
    assert rel(a, fx["theta"]) < 1e-4
    assert rel(one, fx["pose6d"]) > 1e-3                        # ... while iterative_regress itself honours it


def test_dropout_active_in_training_mode(golden):
    reg = build(golden("g12_iterative")).train()
    x = torch.randn(4, 128)
    torch.manual_seed(0)
    a = reg.iterative_regress(x)[0]
    torch.manual_seed(1)
    b = reg.iterative_regress(x)[0]
    assert not torch.allclose(a, b)
