"""Decoder-tail kernels (maed_amd/csrc/smpl.hip forward, tail_bwd.hip backward) on the host simulator: the same
sources that hipcc compiles for gfx950 are compiled for x86 (tests/hostsim) and driven through the product's own
ctypes signatures and autograd Functions (maed_amd/tail.py), then compared with the ATen composition of the same
graph (which tests/test_gpu_model.py ties to the float64 oracle).  Checks arithmetic + wiring without a GPU; the
`-m gpu` suite repeats the comparison on the real library."""
import numpy as np
import os

import pytest
import torch

from maed_amd import tail
from maed_amd.geometry import rot6d_to_rotmat
from maed_amd.ktd import KTD

from _hostsim import option, patched
from maed_amd import _lib as L


def make_ktd(seed=0, feat=48, hidden=32):
    torch.manual_seed(seed)
    ktd = KTD(feat_dim=feat, hidden_dim=hidden).eval()        # eval: Dropout off so both paths see the same graph
    for r in ktd._regressors():
        torch.nn.init.normal_(r.weight, std=0.05)
        torch.nn.init.normal_(r.bias, std=0.3)
    with torch.no_grad():
        ktd.deccam.bias.copy_(torch.tensor([0.9, 0.05, -0.05]))
    return ktd


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def run_both(ktd, x, out_keys, seed=1):
    """-> (reference outputs, reference grads, simulated outputs, simulated grads) for a random cotangent on out_keys"""
    params = list(ktd.parameters())
    out_ref = ktd.get_output(*ktd._head_torch(x), None, hip=False)
    g = torch.Generator().manual_seed(seed)
    cot = {k: torch.randn(out_ref[k].shape, generator=g) for k in out_keys}
    gref = torch.autograd.grad(sum((out_ref[k] * cot[k]).sum() for k in cot), [x] + params, allow_unused=True)
    for p in params:
        p.grad = None
    x.grad = None
    with patched():
        pose, shape, cam = ktd._head_train(x)
        theta, verts, kp2d, kp3d, rotmat = tail.SmplTailFn.apply(pose, shape, cam, ktd.smpl)
        out = dict(theta=theta, verts=verts, kp_2d=kp2d, kp_3d=kp3d, rotmat=rotmat)
        sum((out[k] * cot[k]).sum() for k in cot).backward()
    return out_ref, gref, out, [x.grad] + [p.grad for p in params]


_SLOW = pytest.mark.skipif(os.environ.get("MAED_SLOW_TESTS") != "1", reason="28 s each on the simulator (thread per vertex); the GPU suite runs every subset: MAED_SLOW_TESTS=1")


@pytest.mark.parametrize("out_keys", [("theta", "verts", "kp_2d", "kp_3d", "rotmat"), pytest.param(("kp_2d", "kp_3d", "theta"), marks=_SLOW), ("kp_2d",),
                                      pytest.param(("rotmat",), marks=_SLOW)])
def test_tail_forward_backward_vs_aten(out_keys):
    ktd = make_ktd()
    x = torch.randn(3, 48, requires_grad=True)
    # ("kp_2d",): the skinning kernel's 16-frames-per-workgroup instance (what 128-frame clips take), here with a ragged frame group
    with patched() as lib, option(lib, L.OPT_LBS_FRAMES, 16 if out_keys == ("kp_2d",) else 0):
        out_ref, gref, out, gsim = run_both(ktd, x, out_keys)
    for k in out_ref:
        assert rel(out[k].detach(), out_ref[k].detach()) < 5e-6, k
    names = ["x"] + [n for n, _ in ktd.named_parameters()]
    for n, a, b in zip(names, gsim, gref):
        if b is None:
            assert a is None or float(a.abs().max()) == 0.0, n
            continue
        assert a is not None, n
        assert rel(a, b) < 2e-4, (n, rel(a, b))


def test_head_on_the_split_engine_for_bf16_mode_models():
    """round 4: KTD.head_matmul = "bf16x3" (what MAED sets for a bf16-mode model) -- fc1, fc2, the packed regressor GEMM, their input and weight gradients on the
    split-bf16 MFMA kernels (few output tiles + long K: bias fill + split-K atomics on the matrix cores; the ragged K = 157 input gradient stays exact).  Outputs
    and gradients stay within the split engine's ~2^-16, four orders below what the bf16 encoder in front of the head delivers."""
    ktd = make_ktd(feat=64, hidden=256)
    ktd.head_matmul = "bf16x3"
    x = torch.randn(5, 64, requires_grad=True)
    out_ref, gref, out, gsim = run_both(ktd, x, ("kp_2d", "theta"))
    for k in out_ref:
        assert rel(out[k].detach(), out_ref[k].detach()) < 2e-4, (k, rel(out[k].detach(), out_ref[k].detach()))
    names = ["x"] + [n for n, _ in ktd.named_parameters()]
    for n, a, b in zip(names, gsim, gref):
        if b is None:
            continue
        assert a is not None and rel(a, b) < 2e-3, (n, rel(a, b))


def test_all_quaternion_branches_differentiated():
    """rotation_matrix_to_angle_axis takes one of four branches per joint (geometry.py:143-223); the dual-number backward
    must follow the same branch.  Random 6D poses hit all four; compare d(theta)/d(pose6d) joint by joint."""
    g = torch.Generator().manual_seed(3)
    x6 = torch.randn(40, 144, generator=g)
    R = rot6d_to_rotmat(x6).reshape(-1, 3, 3).transpose(1, 2)
    d2, d0d1, d0nd1 = R[:, 2, 2] < 1e-6, R[:, 0, 0] > R[:, 1, 1], R[:, 0, 0] < -R[:, 1, 1]
    branches = {int(b) for b in (d2 & d0d1) * 0 + (d2 & ~d0d1) * 1 + (~d2 & d0nd1) * 2 + (~d2 & ~d0nd1) * 3}
    assert branches == {0, 1, 2, 3}
    from maed_amd.geometry import rotation_matrix_to_angle_axis
    x_ref = x6.clone().requires_grad_(True)
    Rr = rot6d_to_rotmat(x_ref)
    aa_ref = rotation_matrix_to_angle_axis(Rr).reshape(40, 72)
    cot_aa, cot_R = torch.randn(40, 72, generator=g), torch.randn(40, 24, 9, generator=g)
    ((aa_ref * cot_aa).sum() + (Rr.reshape(40, 24, 9) * cot_R).sum()).backward()
    from maed_amd import _lib as L, ops
    with patched() as lib:
        d_x6 = torch.empty(40, 144)
        d_aa = torch.zeros(40, 85)
        d_aa[:, 3:75] = cot_aa
        L.check(lib.maed_rot6d_pose_bwd(x6.data_ptr(), cot_R.data_ptr(), d_aa.data_ptr() + 12, 85, d_x6.data_ptr(), 40 * 24, None))
    assert rel(d_x6, x_ref.grad) < 1e-4


def test_ktd_pack_unpack_round_trip():
    from maed_amd import _lib as L
    import ctypes as C
    ktd = make_ktd(hidden=32)
    with patched() as lib:
        w_feat, b_feat, w_anc = torch.empty(157, 32), torch.empty(157), torch.empty(L.KTD_W_ANC)
        L.check(lib.maed_ktd_pack(C.byref(ktd._ptr_table(False)), 32, w_feat.data_ptr(), b_feat.data_ptr(), w_anc.data_ptr(), None))
        ref_w = torch.cat([r.weight[:, :32] for r in ktd._regressors()], 0)
        ref_b = torch.cat([r.bias for r in ktd._regressors()], 0)
        ref_a = torch.cat([r.weight[:, 32:].reshape(-1) for r in ktd.joint_regs[1:]], 0)
        assert torch.equal(w_feat, ref_w.detach()) and torch.equal(b_feat, ref_b.detach()) and torch.equal(w_anc, ref_a.detach())
        # unpack_add ACCUMULATES into .grad: run twice, expect 2x
        tbl = ktd._ptr_table(True)
        for _ in range(2):
            L.check(lib.maed_ktd_unpack_add(C.byref(tbl), 32, w_feat.data_ptr(), b_feat.data_ptr(), w_anc.data_ptr(), None))
    for r in ktd._regressors():
        assert torch.allclose(r.weight.grad, 2 * r.weight.detach()) and torch.allclose(r.bias.grad, 2 * r.bias.detach())


def test_grads_ready_protocol_and_fused_parameter_list():
    ktd = make_ktd()
    fired = []
    ktd.grads_ready = lambda m: fired.append(m)
    x = torch.randn(2, 48, requires_grad=True)
    with patched():
        pose, shape, cam = ktd._head_train(x)
        assert ktd._pending_backwards == 1
        (pose.sum() + shape.sum() + cam.sum()).backward()
    assert fired == [ktd] and ktd._pending_backwards == 0
    fused = {id(p) for p in ktd.fused_parameters()}
    assert len(fused) == 52 and all(p.grad is not None for p in ktd.fused_parameters())
    assert id(ktd.fc1.weight) not in fused and ktd.fc1.weight.grad is not None      # fc1/fc2 travel through autograd


def test_forward_without_backward_does_not_block_the_readiness_report():
    """a validation forward on the training model (lib/core/trainer.py validates every epoch, under no_grad) must not leave
    _pending_backwards raised: the next training backward still reports through grads_ready"""
    ktd = make_ktd()
    fired = []
    ktd.grads_ready = lambda m: fired.append(m)
    x = torch.randn(2, 48, requires_grad=True)
    with patched():
        with torch.no_grad():
            ktd._head_train(x)
        assert ktd._pending_backwards == 0
        pose, shape, cam = ktd._head_train(x)
        assert ktd._pending_backwards == 1
        (pose.sum() + shape.sum() + cam.sum()).backward()
    assert fired == [ktd] and ktd._pending_backwards == 0


def test_vertex_backward_on_the_active_vertices_only_matches_the_dense_form(monkeypatch):
    """round 4: without a gradient on the vertices the LBS backward visits only the vertices the extra joints read (maed_smpl_skin_bwd_sparse, ~290 of 6890) and
    contracts only their columns of [posedirs; shapedirs^T]: same parameter / input gradients as the dense kernels (MAED_SMPL_SPARSE_BWD=0), and the dense kernels
    stay in charge when the vertices carry a gradient."""
    ktd = make_ktd()
    act = ktd.smpl.active_vertices()
    assert act.dtype == torch.int32 and 21 <= act.numel() <= 9 * 30 + 21 and bool((act[1:] > act[:-1]).all())
    assert ktd.smpl.pose_shape_dirs_active().shape == (217, 3 * act.numel())
    assert torch.equal(ktd.smpl.pose_shape_dirs_active()[:, 3:6], ktd.smpl.pose_shape_dirs()[:, 3 * int(act[1]):3 * int(act[1]) + 3])
    calls = []
    grads = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MAED_SMPL_SPARSE_BWD", mode)
        x = torch.randn(3, 48, generator=torch.Generator().manual_seed(5)).requires_grad_(True)
        for p in ktd.parameters():
            p.grad = None
        with patched() as lib:
            real = lib.maed_smpl_skin_bwd_sparse
            lib.maed_smpl_skin_bwd_sparse = lambda *a: (calls.append(mode), real(*a))[1]
            try:
                pose, shape, cam = ktd._head_train(x)
                theta, verts, kp2d, kp3d, rotmat = tail.SmplTailFn.apply(pose, shape, cam, ktd.smpl)
                ((kp2d * kp2d).sum() + kp3d.sum()).backward()
            finally:
                lib.maed_smpl_skin_bwd_sparse = real
        grads[mode] = [x.grad.clone()] + [p.grad.clone() for p in ktd.parameters() if p.grad is not None]
    assert calls == ["1"]
    for a, b in zip(grads["1"], grads["0"]):
        assert rel(a, b) < 1e-5, rel(a, b)


def test_joint_regression_through_the_regressors_nonzeros():
    """round 4: maed_joint_regress_csr_fwd -- SMPL's joint regressors are sparse (9 x 30 / 17 x 50 of 6890 columns in the stand-in, like the real ones): CSR built
    once per regressor state, the dense f32-MFMA kernel kept for regressors that are not sparse; both against the einsum"""
    from maed_amd.smpl import SMPL, synthetic_smpl_arrays
    arrays = synthetic_smpl_arrays(0)
    smpl = SMPL(arrays)
    verts = torch.randn(3, 6890, 3, generator=torch.Generator().manual_seed(1))
    with patched() as lib:
        csr = smpl.regressor_csr(smpl.J_regressor_extra)
        assert csr is not None and csr[0].tolist()[0] == 0 and csr[0].tolist()[-1] == csr[1].numel() == csr[2].numel() == int((smpl.J_regressor_extra != 0).sum())
        assert smpl.regressor_csr(smpl.J_regressor_extra) is csr, "cached"
        got = smpl.joint_regress_hip(smpl.J_regressor_extra, verts)
        h36 = smpl.joint_regress_hip(arrays["J_regressor_h36m"], verts)
        dense = torch.rand(5, 6890, generator=torch.Generator().manual_seed(2)) / 6890
        assert smpl.regressor_csr(dense) is None
        gd = smpl.joint_regress_hip(dense, verts)
    for g, R in ((got, smpl.J_regressor_extra), (h36, arrays["J_regressor_h36m"]), (gd, dense)):
        want = torch.einsum("bik,ji->bjk", verts.double(), R.double())
        assert float((g.double() - want).abs().max()) < 1e-5


def test_regressor_csr_cache_follows_the_tensor_not_its_address():
    """the CSR of a joint regressor is cached per tensor object and version: an in-place edit or a different tensor (even one that reuses a freed tensor's storage
    address) gets its own"""
    from maed_amd.smpl import SMPL, synthetic_smpl_arrays
    smpl = SMPL(synthetic_smpl_arrays(0))
    a = torch.zeros(2, 6890); a[0, 5] = 1.0; a[1, 7] = 2.0
    c1 = smpl.regressor_csr(a)
    assert c1[1].tolist() == [5, 7] and smpl.regressor_csr(a) is c1
    a[1, 9] = 3.0                                    # in-place edit: version bump
    c2 = smpl.regressor_csr(a)
    assert c2 is not c1 and c2[1].tolist() == [5, 7, 9]
    ptr = a.data_ptr()
    del a
    b = torch.zeros(2, 6890); b[0, 1] = 1.0          # (often lands on the freed storage)
    c3 = smpl.regressor_csr(b)
    assert c3[1].tolist() == [1], (ptr == b.data_ptr(), c3[1].tolist())


def test_lbs_blend_on_the_matrix_cores_matches_the_valu_kernel():
    """round 5: maed_smpl_lbs_fwd's default route -- v_posed = [pose features | betas | 1] . [posedirs; shapedirs^T; v_template] on v_mfma_f32_32x32x2_f32 (a wave
    per 32-column tile and 64 frames; the last column tile is partial: 20670 = 645 * 32 + 30) followed by the streaming skinning pass -- against the VALU kernel
    (MAED_OPT_LBS_FRAMES = 4) and the fp64 ATen composition: a partial frame group (3 of 64 frames), in place (inference: no v_posed buffer) and with the buffer
    the training path keeps for the backward."""
    import ctypes as C
    from _hostsim import option
    from maed_amd import _lib as L, ops
    from maed_amd.smpl import SMPL
    torch.manual_seed(4)
    smpl = SMPL()
    Fr = 3
    betas = torch.randn(Fr, 10)
    from maed_amd.geometry import rot6d_to_rotmat
    rot = rot6d_to_rotmat(torch.randn(Fr * 24, 6)).reshape(Fr, 24, 3, 3).contiguous()
    res = {}
    with patched() as lib:
        for mode in (4, 0):
            with option(lib, L.OPT_LBS_FRAMES, mode):
                res[mode] = [t.clone() for t in smpl.lbs_hip(betas, rot)]
        with option(lib, L.OPT_LBS_FRAMES, 0):      # with the v_posed buffer (training)
            verts = torch.empty(Fr, 6890, 3); j24 = torch.empty(Fr, 24, 3); A = torch.empty(Fr, 24, 12); vp = torch.empty(Fr, 6890, 3)
            sp = smpl._c_params()
            ops.check(lib.maed_smpl_lbs_fwd(C.byref(sp), ops._p(betas), ops._p(rot), ops._p(verts), ops._p(j24), ops._p(A), ops._p(vp), Fr, None), "smpl_lbs_fwd")
    scale = res[4][0].abs().max()
    assert (res[0][0] - res[4][0]).abs().max() <= 2e-6 * scale        # same fp32 products, another summation order
    assert torch.equal(res[0][1], res[4][1])
    assert torch.equal(verts, res[0][0])
    # v_posed itself against the closed form in fp64
    feat = (rot[:, 1:] - torch.eye(3)).reshape(Fr, 207).double()
    vp_ref = (smpl.v_template.double().reshape(1, -1) + betas.double() @ smpl.shapedirs.double().reshape(-1, 10).t() + feat @ smpl.posedirs.double()).reshape(Fr, 6890, 3)
    assert (vp.double() - vp_ref).abs().max() <= 2e-6 * vp_ref.abs().max()
