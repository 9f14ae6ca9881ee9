"""Backbone helper kernels (maed_amd/csrc/backbone.hip) on the host simulator against ATen on CPU: SAME max-pool forward /
gather backward (incl. ATen's tie rule on ReLU zeros), fused GroupNorm(+residual)(+ReLU) forward/backward, batched weight
standardisation forward/backward -- through the product's own autograd Functions (maed_amd/ops.py)."""
import pytest
import torch
import torch.nn.functional as F

from maed_amd import ops

from _hostsim import patched


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("N,C,H,W", [(2, 8, 8, 8), (1, 16, 7, 9), (2, 8, 12, 5)])
def test_maxpool_same_matches_aten_with_ties(N, C, H, W):
    torch.manual_seed(0)
    x = F.relu(torch.randn(N, C, H, W))                 # many exact zeros: ties inside windows, as after the stem's ReLU
    x[0, :, 0, 0] = float("nan") if False else x[0, :, 0, 0]
    xr = x.clone().requires_grad_(True)
    k, s = 3, 2
    ph = max((-(-H // s) - 1) * s + k - H, 0); pw = max((-(-W // s) - 1) * s + k - W, 0)
    ref = F.max_pool2d(F.pad(xr, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2], value=-float("inf")), k, s, 0)
    g = torch.randn_like(ref)
    ref.backward(g)
    xs = cl(x.clone()).requires_grad_(True)
    with patched():
        y = ops.MaxPool3s2SameFn.apply(xs)
        y.backward(cl(g))
    assert torch.equal(y.detach(), ref.detach())
    assert torch.allclose(xs.grad, xr.grad, atol=1e-6), (xs.grad - xr.grad).abs().max()


@pytest.mark.parametrize("defer_affine", ["0", "1"])      # MAED_GN_DEFER_AFFINE: dgamma/dbeta from the per-frame partials instead of atomics
@pytest.mark.parametrize("relu,res", [(True, True), (True, False), (False, False), (False, True)])
def test_groupnorm_fused_matches_aten(relu, res, defer_affine, monkeypatch):
    monkeypatch.setenv("MAED_GN_DEFER_AFFINE", defer_affine)
    torch.manual_seed(1)
    N, C, H, W = 2, 64, 5, 6
    x = torch.randn(N, C, H, W); r = torch.randn(N, C, H, W) if res else None
    gamma, beta = torch.randn(C), torch.randn(C)
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if res else None
    ref = F.group_norm(xr, 32, gr, br, 1e-5)
    if res:
        ref = ref + rr
    if relu:
        ref = F.relu(ref)
    g = torch.randn_like(ref)
    ref.backward(g)
    xs = cl(x.clone()).requires_grad_(True)
    rs = cl(r.clone()).requires_grad_(True) if res else None
    gs, bs = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    with patched():
        y = ops.GroupNormFn.apply(xs, rs, gs, bs, 1e-5, relu, False)
        y.backward(cl(g))
    assert torch.allclose(y.detach(), ref.detach(), rtol=1e-4, atol=1e-4)
    assert torch.allclose(xs.grad, xr.grad, rtol=1e-3, atol=1e-4)
    assert torch.allclose(gs.grad, gr.grad, rtol=1e-3, atol=1e-3) and torch.allclose(bs.grad, br.grad, rtol=1e-3, atol=1e-3)
    if res:
        assert torch.allclose(rs.grad, rr.grad, rtol=1e-4, atol=1e-5)


def test_weight_standardisation_batched_matches_aten():
    torch.manual_seed(2)

    class Owner:
        _pending_backwards = 0
        grads_ready = None

        def __init__(self, ws):
            self._ws = ws

        def conv_weights(self):
            return self._ws
    shapes = [(8, 3, 7, 7), (16, 8, 1, 1), (8, 16, 3, 3)]
    ws = [torch.randn(s, requires_grad=True) for s in shapes]
    refs = [w.detach().clone().requires_grad_(True) for w in ws]
    owner = Owner(ws)
    fired = []
    owner.grads_ready = lambda o: fired.append(o)
    cots = [torch.randn(s) for s in shapes]
    outs_ref = []
    for w in refs:
        std, mean = torch.std_mean(w, dim=[1, 2, 3], keepdim=True, unbiased=False)
        outs_ref.append((w - mean) / (std + 1e-5))
    sum((o * c).sum() for o, c in zip(outs_ref, cots)).backward()
    with patched():
        outs = ops.WeightStdFn.apply(owner, torch.float32, 1e-5, *ws)
        sum((o * c).sum() for o, c in zip(outs, cots)).backward()
    assert fired == [owner]
    for o, r, w, wr in zip(outs, outs_ref, ws, refs):
        assert torch.allclose(o.detach(), r.detach(), rtol=1e-4, atol=1e-5)
        assert torch.allclose(w.grad, wr.grad, rtol=1e-3, atol=1e-4), (w.grad - wr.grad).abs().max()
