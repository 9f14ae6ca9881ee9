"""Round 2: every differentiable piece that used to detour through ATen on the device now runs on libmaed_hip behind autograd Functions --
stand-alone Attention / Mlp in 'parallel' mode, the final LayerNorm + pre_logits tanh of the training graph, KTD's fc1 / fc2 + Dropout,
SMPL.forward with gradients, the acceleration loss.  Kernels on the host simulator, against fp64 autograd through the CPU oracle / the
module's own ATen composition (itself pinned to the reference on CPU)."""
import numpy as np
import pytest
import torch

from oracle import maed_ref as R
from maed_amd import _lib as L
from maed_amd import ops, ste_modes
from maed_amd.vision_transformer import Attention, Mlp

from _hostsim import patched
from _util import rnd


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def test_standalone_attention_and_mlp_parallel_mode_are_differentiable(golden):
    """vision_transformer.py:136-178 Attention.forward / :106-112 Mlp.forward used on their own WITH gradients (VERDICT r1 missing 5):
    forward against the reference's stored output (g1, g3), every gradient against fp64 autograd through the oracle"""
    g1 = golden("g1_attention")
    H, T = int(g1["heads"]), int(g1["seqlen"])
    t = lambda a: torch.from_numpy(np.asarray(a))
    state = {k[3:]: t(g1[k]) for k in g1.files if k.startswith("sd.")}
    att = Attention(128, num_heads=H, qkv_bias=True, st_mode="parallel")
    att.load_state_dict(state)
    x = t(g1["x"]).clone().requires_grad_(True)
    cot = rnd(*x.shape, seed=3)
    pd = {k: v.double().requires_grad_(True) for k, v in state.items()}
    xr = x.detach().double().requires_grad_(True)
    ref = R.attention_parallel(xr, pd, "", H, T)
    (ref * cot.double()).sum().backward()
    with patched():
        out = att(x, T, compute_dtype=torch.float32)
        (out * cot).sum().backward()
    assert rel(out, t(g1["out"])) < 2e-5 and rel(out, ref) < 2e-5
    assert rel(x.grad, xr.grad) < 1e-4
    for n, p in att.named_parameters():
        assert p.grad is not None and rel(p.grad, pd[n].grad) < 2e-4, (n, rel(p.grad, pd[n].grad))
    g3 = golden("g3_mlp_ln")
    m = Mlp(128, 512)
    ms = {k[4:]: t(g3[k]) for k in g3.files if k.startswith("mlp.") and k != "mlp_out"}
    m.load_state_dict(ms)
    xm = t(g3["x"]).clone().requires_grad_(True)
    cm = rnd(*g3["mlp_out"].shape, seed=4)
    pm = {k: v.double().requires_grad_(True) for k, v in ms.items()}
    xmr = xm.detach().double().requires_grad_(True)
    refm = R.mlp(xmr, pm, "")
    (refm * cm.double()).sum().backward()
    with patched():
        y = m(xm)
        (y * cm).sum().backward()
    assert rel(y, t(g3["mlp_out"])) < 2e-5 and rel(xm.grad, xmr.grad) < 1e-4
    for n, p in m.named_parameters():
        assert rel(p.grad, pm[n].grad) < 2e-4, n


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_training_tail_of_the_encoder_layernorm_tanh_linear(dtype):
    """final LayerNorm on the cls rows + pre_logits Linear + tanh (vision_transformer.py:344,350-353,404-407) with gradients"""
    Fr, P, C = 6, 5, 128
    g = torch.Generator().manual_seed(5)
    tok = torch.randn(Fr, P, C, generator=g).requires_grad_(True)
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_(True), (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    W, b = (torch.randn(C, C, generator=g) * C ** -0.5).requires_grad_(True), (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    cot = torch.randn(Fr, C, generator=g)
    leaves = [tok, gamma, beta, W, b]
    ref_leaves = [v.detach().double().requires_grad_(True) for v in leaves]
    rt, rg, rb, rW, rbb = ref_leaves
    ref = torch.tanh(torch.nn.functional.linear(R.layer_norm(rt[:, 0], rg, rb), rW, rbb))
    (ref * cot.double()).sum().backward()
    cache = ops.WeightCache()
    with patched():
        y = ste_modes.LayerNormFn.apply(tok[:, 0], gamma, beta, 1e-6, dtype)
        out = ste_modes.TanhLinearFn.apply(y, W, b, cache).float()
        (out * cot).sum().backward()
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    assert rel(out, ref) < tol
    for a, r, n in zip(leaves, ref_leaves, ["tok", "gamma", "beta", "W", "b"]):
        assert rel(a.grad, r.grad) < (2e-4 if dtype == torch.float32 else 5e-2), (n, rel(a.grad, r.grad))
    assert torch.count_nonzero(tok.grad[:, 1:]) == 0              # only the cls rows receive a gradient


def test_dropout_kernel_distribution_scaling_and_backward_mask():
    """nn.Dropout(p) semantics (ktd.py:54,56): keep probability 1 - p, survivors scaled by 1 / (1 - p), backward through the SAME mask;
    identity in eval; a different seed draws a different mask, the same seed the same one"""
    x = torch.ones(64, 1024)
    with patched():
        torch.manual_seed(3)
        xg = x.clone().requires_grad_(True)
        y = ste_modes.dropout(xg, 0.5, True)
        y.sum().backward()
        torch.manual_seed(3)
        y_same = ste_modes.dropout(x, 0.5, True)
        y_other = ste_modes.dropout(x, 0.5, True)
        y25 = ste_modes.dropout(x, 0.25, True)
        assert ste_modes.dropout(x, 0.5, False) is x
    kept = (y != 0)
    assert set(y.unique().tolist()) == {0.0, 2.0} and abs(kept.float().mean().item() - 0.5) < 0.01
    assert torch.equal(xg.grad, y)                                 # d/dx of sum(y) = mask / (1 - p)
    assert torch.equal(y, y_same) and not torch.equal(y, y_other)
    assert abs((y25 != 0).float().mean().item() - 0.75) < 0.01 and abs(y25.max().item() - 1 / 0.75) < 1e-6
    # no structure along rows / columns (a counter-based hash, not a per-row pattern)
    assert kept.float().mean(0).std().item() < 0.08 and kept.float().mean(1).std().item() < 0.03


def test_dropout_with_the_seed_in_the_device_record():
    """maed_dropout_dev (round 6): the mask of Dropout layer `call_id` is maed_dropout's with seed + call_id * 0x9E3779B97F4A7C15; the backward re-draws the forward's
    mask; a new step (begin_step) draws new masks, the same seed the same ones"""
    from maed_amd import _lib as L
    x = torch.ones(32, 512)
    st = ops.DeviceTrainState(torch.device("cpu"))
    prev = ops.DEVICE_STATE
    with patched() as lib:
        try:
            ops.DEVICE_STATE = st
            st.begin_step(1234); st.upload()
            xg = x.clone().requires_grad_(True)
            y1 = ste_modes.dropout(xg, 0.5, True)          # call_id 1
            y2 = ste_modes.dropout(x, 0.5, True)           # call_id 2
            y1.sum().backward()
            st.begin_step(1234); st.upload()
            y1_again = ste_modes.dropout(x, 0.5, True)
            st.begin_step(99); st.upload()
            y1_other = ste_modes.dropout(x, 0.5, True)
        finally:
            ops.DEVICE_STATE = prev
        ref = torch.empty_like(x)
        seed = (1234 + 1 * 0x9E3779B97F4A7C15) % (1 << 64)
        L.check(lib.maed_dropout(x.data_ptr(), ref.data_ptr(), x.numel(), 0.5, seed, None), "dropout")
    assert torch.equal(y1.detach(), ref) and torch.equal(xg.grad, ref)
    assert torch.equal(y1.detach(), y1_again) and not torch.equal(y1.detach(), y2) and not torch.equal(y1.detach(), y1_other)
    assert abs((y2 != 0).float().mean().item() - 0.5) < 0.02


def test_smpl_forward_with_gradients_runs_on_the_library():
    """SMPL.forward (lib/models/smpl.py:94-106) with grad: tail.SmplLbsFn against the module's ATen composition (CPU path)"""
    from maed_amd.smpl import SMPL, synthetic_smpl_arrays
    from maed_amd.geometry import rot6d_to_rotmat
    smpl = SMPL(synthetic_smpl_arrays(0))
    g = torch.Generator().manual_seed(2)
    Fr = 3
    betas = torch.randn(Fr, 10, generator=g)
    rot = rot6d_to_rotmat(torch.randn(Fr * 24, 6, generator=g)).reshape(Fr, 24, 3, 3)
    cv, cj = torch.randn(Fr, 6890, 3, generator=g), torch.randn(Fr, 49, 3, generator=g)
    outs = {}
    for name in ("aten", "lib"):
        b, r = betas.clone().requires_grad_(True), rot.clone().requires_grad_(True)
        if name == "lib":
            with patched():
                o = smpl(betas=b, body_pose=r[:, 1:], global_orient=r[:, :1], pose2rot=False)
                ((o.vertices * cv).sum() + (o.joints * cj).sum()).backward()
        else:
            o = smpl(betas=b, body_pose=r[:, 1:], global_orient=r[:, :1], pose2rot=False)
            ((o.vertices * cv).sum() + (o.joints * cj).sum()).backward()
        outs[name] = (o.vertices.detach(), o.joints.detach(), b.grad, r.grad)
    for a, b_, n in zip(outs["lib"], outs["aten"], ["verts", "joints", "d_betas", "d_rotmat"]):
        assert rel(a, b_) < 2e-4, (n, rel(a, b_))


def test_ktd_refuses_a_differentiable_graph_with_an_evaluation_regressor():
    from test_hostsim_tail import make_ktd
    ktd = make_ktd()
    x = torch.randn(2, 48, requires_grad=True)
    with patched():
        with pytest.raises(NotImplementedError):
            ktd(x, 1, J_regressor=torch.rand(17, 6890))
        with torch.no_grad():
            out = ktd.eval()(x, 1, J_regressor=torch.rand(17, 6890))
    assert out["kp_3d"].shape == (2, 17, 3)


def test_parameter_gradients_written_straight_into_grad_equal_autograds():
    """ops.direct_grad_slot (round 6): at the single-use call sites (KTD fc1 / fc2, pre_logits, the encoder's final LayerNorm) the backward kernels accumulate into
    an existing fp32 .grad themselves -- no zero-fill, no AccumulateGrad add -- and report through the hook the bucketer left on the parameter.  Same numbers as
    autograd's way (MAED_DIRECT_GRADS = 0), accumulation into a non-zero .grad, two forwards before one backward report once, after the second backward."""
    from maed_amd.ste_modes import LinearTokFn, LayerNormFn, TanhLinearFn
    from maed_amd.ops import WeightCache
    torch.manual_seed(0)
    lin, lin2, ln = torch.nn.Linear(64, 32), torch.nn.Linear(32, 16), torch.nn.LayerNorm(64, eps=1e-6)
    x1, x2 = torch.randn(40, 64), torch.randn(24, 64)
    params = list(lin.parameters()) + list(lin2.parameters()) + list(ln.parameters())
    reported = []

    def run(direct, preset):
        for p in params:
            p.grad = torch.full_like(p, preset) if preset is not None else None
            p._maed_ready = reported.append
            p._maed_direct = 0
        reported.clear()
        c1, c2 = WeightCache(), WeightCache()
        out = 0
        for x in (x1, x2):          # two forwards, one backward (trainer.py:253-262)
            h = LayerNormFn.apply(x, ln.weight, ln.bias, ln.eps, torch.float32, True)
            h = LinearTokFn.apply(h, lin.weight, lin.bias, c1, False, "bf16x3", True)          # fp32 rows on the split-bf16 engine (the decoder head's mode)
            out = out + TanhLinearFn.apply(h.bfloat16(), lin2.weight, lin2.bias, c2, True).float().square().sum()
        out.backward()
        return [p.grad.clone() for p in params], list(reported)

    with patched():
        old = ops.DIRECT_GRADS
        try:
            ops.DIRECT_GRADS = False
            ref, rep0 = run(False, 0.25)
            ops.DIRECT_GRADS = True
            got, rep1 = run(True, 0.25)
            none_grads, rep2 = run(True, None)          # no .grad to accumulate into: autograd's way, nothing reported by the kernels
        finally:
            ops.DIRECT_GRADS = old
    assert rep0 == [] and rep2 == []
    assert len(rep1) == len(params) and {id(p) for p in rep1} == {id(p) for p in params}      # each parameter once, after its LAST backward
    for a, b, c in zip(ref, got, none_grads):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-4)
        assert torch.allclose(a - 0.25, c, rtol=1e-5, atol=1e-4)
    assert all(getattr(p, "_maed_direct", 0) == 0 for p in params)
