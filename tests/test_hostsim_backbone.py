"""Backbone helper kernels (maed_amd/csrc/backbone.hip) on the host simulator against ATen on CPU: SAME max-pool forward /
gather backward (incl. ATen's tie rule on ReLU zeros), fused GroupNorm(+residual)(+ReLU) forward/backward, batched weight
standardisation forward/backward -- through the product's own autograd Functions (maed_amd/ops.py)."""
import os

import pytest
import torch
import torch.nn.functional as F

from maed_amd import _lib as L
from maed_amd import ops

from _hostsim import option, patched


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


@pytest.mark.parametrize("relu,res", [(True, True), (True, False), (False, False), (False, True)])
def test_groupnorm_fused_matches_aten(relu, res):
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


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("relu,res", [(True, True), (True, False), (False, False)])
@pytest.mark.parametrize("N,C,H,W", [(2, 64, 20, 21), (1, 128, 10, 13), (2, 256, 14, 14), (1, 1024, 6, 6)])
def test_groupnorm_backward_one_pass_register_resident(dtype, relu, res, N, C, H, W):
    """round 4: maed_groupnorm_bwd with frame_sync -- x and dy are read ONCE, the slices stay in registers between the reduction and the apply step, the workgroups
    of a frame meet at a per-frame counter.  Every shape here splits a frame over several workgroups (2-4 slices, ragged last slice, 1 / 2 / 4 groups per
    8-channel chunk); the simulator runs workgroups one after another, so its launcher issues the two halves of the kernel as two launches: everything but the spin
    itself is checked here -- against fp64 autograd on the same (rounded) inputs, and against the two-pass kernels."""
    torch.manual_seed(5)
    x = torch.randn(N, C, H, W).to(dtype); r = torch.randn(N, C, H, W).to(dtype) if res else None
    gamma, beta = torch.randn(C), torch.randn(C)
    g = torch.randn(N, C, H, W).to(dtype)
    xr, gr, br = x.double().requires_grad_(True), gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rr = r.double().requires_grad_(True) if res else None
    ref = F.group_norm(xr, 32, gr, br, 1e-5)
    if res:
        ref = ref + rr
    if relu:
        ref = F.relu(ref)
    ref.backward(g.double())
    out = {}
    for onepass in (1, 0):
        xs = cl(x.clone()).requires_grad_(True)
        rs = cl(r.clone()).requires_grad_(True) if res else None
        gs, bs = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        old = L.get_option(L.OPT_GN_BWD_ONEPASS)
        with patched():
            L.set_option(L.OPT_GN_BWD_ONEPASS, onepass)
            try:
                y = ops.GroupNormFn.apply(xs, rs, gs, bs, 1e-5, relu, False)
                y.backward(cl(g))
            finally:
                L.set_option(L.OPT_GN_BWD_ONEPASS, old)
        out[onepass] = (xs.grad.double(), gs.grad.double(), bs.grad.double(), rs.grad.double() if res else None)
    tol = 2e-5 if dtype == torch.float32 else 1.2e-2
    for k, (dx, dg, db, dr) in out.items():
        # near a ReLU kink a bf16-rounded output may flip sign against the fp64 reference: compare where the reference is clear of zero
        clear = (ref.detach().abs() > (0 if dtype == torch.float32 else 2e-2)) | (not relu)
        assert ((dx - xr.grad).abs() * clear).max() <= tol * xr.grad.abs().max(), (k, "dx")
        if dtype == torch.float32:
            assert (dg - gr.grad).abs().max() <= 1e-4 * gr.grad.abs().max() and (db - br.grad).abs().max() <= 1e-4 * br.grad.abs().max(), (k, "affine")
        if res and dtype == torch.float32:
            assert (dr - rr.grad).abs().max() <= 1e-6 * rr.grad.abs().max(), (k, "dres")
    # one pass vs two passes: the same arithmetic up to summation order
    assert (out[1][0] - out[0][0]).abs().max() <= (1e-5 if dtype == torch.float32 else 8e-3) * out[0][0].abs().max()
    assert (out[1][1] - out[0][1]).abs().max() <= 1e-4 * out[0][1].abs().max() and (out[1][2] - out[0][2]).abs().max() <= 1e-4 * out[0][2].abs().max()
    if res:
        assert torch.equal(out[1][3], out[0][3])


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


def test_weight_standardisation_direct_convs_get_transposed_images_and_fp32_slices():
    """bf16 mode, convolutions on the library's own kernels (owner._direct_convs: 1x1 GEMM convolutions and, with MAED_CONV3X3=own, the
    stride-1 3x3 ones): ws_fwd also writes the (kh,kw,I,O) transposed image, ws_bwd reads their weight gradient from an fp32 (O, kh*kw*I) slice"""
    torch.manual_seed(3)

    class Owner:
        _pending_backwards = 0
        grads_ready = None
        _direct_convs = [1, 2]

        def __init__(self, ws):
            self._ws = ws

        def conv_weights(self):
            return self._ws
    shapes = [(8, 3, 7, 7), (16, 8, 1, 1), (8, 16, 3, 3)]
    ws = [torch.randn(s, requires_grad=True) for s in shapes]
    refs = [w.detach().clone().requires_grad_(True) for w in ws]
    owner = Owner(ws)
    cots = [torch.randn(s) for s in shapes]
    outs_ref = []
    for w in refs:
        std, mean = torch.std_mean(w, dim=[1, 2, 3], keepdim=True, unbiased=False)
        outs_ref.append((w - mean) / (std + 1e-5))
    sum((o * c).sum() for o, c in zip(outs_ref, cots)).backward()
    with patched():
        outs = ops.WeightStdFn.apply(owner, torch.bfloat16, 1e-5, *ws)
        for i in (1, 2):
            O, I, kh, kw = shapes[i]
            wt = owner._w_std_t[i]
            assert wt.shape == (kh * kw * I, O) and owner._dw_slices[i].shape == (O, kh * kw * I)
            assert torch.allclose(wt.float(), outs_ref[i].detach().permute(2, 3, 1, 0).reshape(kh * kw * I, O), rtol=1e-2, atol=1e-2)
            owner._dw_slices[i].copy_(cots[i].permute(0, 2, 3, 1).reshape(O, -1))          # what Conv1x1Fn / Conv3x3Fn leave there
        (outs[0].float() * cots[0]).sum().backward()                                        # conv 0 through autograd (bf16 gradient)
    for i, (o, r) in enumerate(zip(outs, outs_ref)):
        assert torch.allclose(o.float(), r.detach(), rtol=1e-2, atol=1e-2)
    for i in (1, 2):
        assert torch.allclose(ws[i].grad, refs[i].grad, rtol=1e-3, atol=1e-4), (i, (ws[i].grad - refs[i].grad).abs().max())
    assert torch.allclose(ws[0].grad, refs[0].grad, rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("N,I,O,H,W,stride", [(2, 64, 64, 6, 5, 1), (1, 128, 72, 7, 9, 1), (2, 64, 136, 8, 8, 2), (1, 64, 64, 7, 7, 2), (1, 64, 64, 14, 14, 1),
                                              (2, 64, 64, 8, 4, 1), (4, 128, 64, 4, 8, 1), (1, 64, 136, 8, 16, 1), (2, 64, 64, 16, 12, 1)])
def test_conv3x3_implicit_gemm_matches_aten(N, I, O, H, W, stride, monkeypatch):
    """maed_conv3x3_fwd (gathered A rows, zero page for out-of-image taps) and Conv3x3Fn's input gradient against F.conv2d with the
    reference's TF-SAME padding (resnetv2.py:51-59: pad//2 in front), odd and even sizes, stride 1 and 2, ragged row / column tiles"""
    from maed_amd.resnetv2 import _same_pad
    own_wgrad = []
    real = ops.conv3x3_wgrad
    monkeypatch.setattr(ops, "conv3x3_wgrad", lambda dy_, x_, **k: (own_wgrad.append(1), real(dy_, x_, **k))[1])
    g = torch.Generator().manual_seed(H * 100 + W)
    bf = torch.bfloat16
    x = torch.randn(N, I, H, W, generator=g).to(bf).float()
    w = (torch.randn(O, I, 3, 3, generator=g) * (1.0 / (3 * I ** 0.5))).to(bf).float()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv2d(_same_pad(xr, 3, stride), wr, None, stride)
    dy = torch.randn(ref.shape, generator=g).to(bf).float()
    ref.backward(dy)
    xs = cl(x.to(bf)).requires_grad_(True)
    ws = w.to(bf).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).requires_grad_(True)       # (O,I,3,3) over (O,3,3,I) storage
    with patched():
        y = ops.Conv3x3Fn.apply(xs, ws, stride)
        assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
        y.backward(cl(dy.to(bf)))
    assert torch.allclose(y.float(), ref.detach(), rtol=2e-2, atol=2e-2), (y.float() - ref).abs().max()
    assert torch.allclose(xs.grad.float(), xr.grad, rtol=2e-2, atol=2e-2), (xs.grad.float() - xr.grad).abs().max()
    assert torch.allclose(ws.grad.float(), wr.grad, rtol=3e-2, atol=3e-2 * float(wr.grad.abs().max()))
    # the library's own weight-gradient kernel (TN GEMM over gathered rows + tap mask) runs for stride 1 when F*H*W % 64 == 0
    assert bool(own_wgrad) == (stride == 1 and (N * H * W) % 64 == 0)


@pytest.mark.parametrize("N,I,O,H,W", [(2, 64, 64, 8, 4), (1, 128, 64, 8, 16), (1, 64, 128, 6, 5)])
def test_conv3x3_with_transposed_image_and_direct_dw_slice(N, I, O, H, W):
    """the hand-over used inside the backbone: input gradient read in place from the (3,3,I,O) transposed image WeightStdFn writes
    (tap flip = negative tap stride), weight gradient accumulated into an fp32 (O, 9*I) slice (also when it falls back to the framework
    because F*H*W is not a multiple of 64)"""
    g = torch.Generator().manual_seed(N * 10 + W)
    bf = torch.bfloat16
    x = torch.randn(N, I, H, W, generator=g).to(bf).float()
    w = (torch.randn(O, I, 3, 3, generator=g) * (1.0 / (3 * I ** 0.5))).to(bf).float()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, 1, 1)
    dy = torch.randn(ref.shape, generator=g).to(bf).float()
    ref.backward(dy)
    xs = cl(x.to(bf)).requires_grad_(True)
    ws = w.to(bf).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)                 # no grad: the slice carries dW
    wt = w.to(bf).permute(2, 3, 1, 0).contiguous()                                     # (3,3,I,O): what ws_fwd's dst_t image holds
    prior = torch.randn(O, 9 * I, generator=g)
    dw = prior.clone()
    with patched():
        y = ops.Conv3x3Fn.apply(xs, ws, 1, wt, dw)
        y.backward(cl(dy.to(bf)))
    assert torch.allclose(y.float(), ref.detach(), rtol=2e-2, atol=2e-2)
    assert torch.allclose(xs.grad.float(), xr.grad, rtol=2e-2, atol=2e-2), (xs.grad.float() - xr.grad).abs().max()
    got = (dw - prior).view(O, 3, 3, I).permute(0, 3, 1, 2)
    assert torch.allclose(got, wr.grad, rtol=3e-2, atol=3e-2 * float(wr.grad.abs().max())), (got - wr.grad).abs().max()


@pytest.mark.parametrize("F,HW,Cin,Cout", [(3, 144, 64, 64), (2, 200, 128, 256), (1, 130, 64, 128)])
def test_conv1x1_epilogue_accumulates_groupnorm_statistics(F, HW, Cin, Cout):
    """maed_conv1x1_fwd gn_sums: (sum, sum of squares) per (frame, group) of the STORED bf16 output, for 2 / 8 / 4 channels per group, frames
    that straddle 128-row tiles (144, 200, 130 rows per frame) and a ragged last tile"""
    torch.manual_seed(0)
    x = torch.randn(F, Cin, HW, 1).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 1, 1) * Cin ** -0.5).bfloat16()
    sums = torch.zeros(F, 32, 2, dtype=torch.float64)
    with patched():
        y = ops.Conv1x1Fn.apply(x, w, None, None, False, sums)
        y0 = ops.Conv1x1Fn.apply(x, w, None, None, False, None)
    assert torch.equal(y, y0)
    yg = y.float().permute(0, 2, 3, 1).reshape(F, HW, 32, Cout // 32).double()
    want = torch.stack([yg.sum(dim=(1, 3)), (yg * yg).sum(dim=(1, 3))], dim=-1)
    assert torch.allclose(sums, want, rtol=1e-5, atol=1e-3), (sums - want).abs().max()


@pytest.mark.parametrize("F,H,W,Cin,Cout,stride", [(2, 12, 12, 64, 64, 1), (1, 23, 12, 64, 128, 1), (2, 24, 24, 128, 128, 2)])
def test_conv3x3_epilogue_accumulates_groupnorm_statistics(F, H, W, Cin, Cout, stride):
    torch.manual_seed(1)
    x = torch.randn(F, Cin, H, W).bfloat16().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(Cout, 3, 3, Cin) * (9 * Cin) ** -0.5).bfloat16()
    Ho, Wo = -(-H // stride), -(-W // stride)
    sums = torch.zeros(F, 32, 2, dtype=torch.float64)
    with patched():
        y = ops.conv3x3(x, wt, stride, gn_sums=sums)
        y0 = ops.conv3x3(x, wt, stride)
    assert torch.equal(y, y0)
    yg = y.float().permute(0, 2, 3, 1).reshape(F, Ho * Wo, 32, Cout // 32).double()
    want = torch.stack([yg.sum(dim=(1, 3)), (yg * yg).sum(dim=(1, 3))], dim=-1)
    assert torch.allclose(sums, want, rtol=1e-5, atol=1e-3), (sums - want).abs().max()


def test_groupnorm_statistics_shape_guard_is_loud():
    x = torch.randn(1, 64, 8, 8).bfloat16().contiguous(memory_format=torch.channels_last)      # 64 pixels per frame < 128
    w = torch.randn(64, 64, 1, 1).bfloat16()
    with patched(), pytest.raises(RuntimeError, match="GroupNorm statistics"):
        ops.Conv1x1Fn.apply(x, w, None, None, False, torch.zeros(1, 32, 2, dtype=torch.float64))


@pytest.mark.parametrize("N,I,O,H,W", [(2, 64, 128, 6, 8), (1, 128, 64, 7, 5)])
def test_conv1x1_stride2_on_packed_pixels_matches_aten(N, I, O, H, W):
    """the downsample shortcut of stages 2 / 3 (resnetv2.py:207-216): Conv1x1Fn(stride=2) = maed_subsample2_fwd + GEMMs on a quarter of the rows +
    maed_subsample2_bwd, against F.conv2d(stride=2) -- even and odd extents"""
    torch.manual_seed(3)
    x = torch.randn(N, I, H, W).bfloat16()
    w = (torch.randn(O, I, 1, 1) * I ** -0.5).bfloat16()
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, stride=2)
    dy = torch.randn(*ref.shape).bfloat16()
    ref.backward(dy.double())
    xs = x.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wt = w.reshape(O, I).t().contiguous()
    dw = torch.zeros(O, I)
    with patched():
        y = ops.Conv1x1Fn.apply(xs, w, wt, dw, False, None, 2)
        y.backward(dy)
    assert y.shape == ref.shape and torch.allclose(y.double(), ref, rtol=2e-2, atol=2e-2)
    assert torch.allclose(xs.grad.double(), xr.grad, rtol=2e-2, atol=2e-2)
    assert (xs.grad[:, :, 1::2] == 0).all() and (xs.grad[:, :, :, 1::2] == 0).all()
    assert torch.allclose(dw.double(), wr.grad.reshape(O, I), rtol=1e-3, atol=1e-3 * wr.grad.abs().max().item())


@pytest.mark.parametrize("N,I,O,H,W", [(2, 64, 64, 8, 8), (1, 128, 64, 7, 9), (1, 64, 128, 6, 5), (1, 64, 64, 14, 14)])
def test_conv3x3_stride2_input_gradient_by_parity_classes(N, I, O, H, W):
    """round 4: maed_conv3x3_s2_dgrad -- the input gradient of the stride-2 3x3 SAME convolution as four implicit GEMMs, one per parity class of the input
    pixel (2 or 1 forward taps per axis), against autograd through F.conv2d on the TF-SAME padded input: even extents (pad 0 top / left, 1 bottom / right) and odd
    ones (1 / 1), every pixel of dx written exactly once (the output starts as NaN)."""
    import math
    torch.manual_seed(7)
    x = torch.randn(N, I, H, W).bfloat16()
    w = (torch.randn(O, I, 3, 3) * (9 * I) ** -0.5).bfloat16()
    Ho, Wo = math.ceil(H / 2), math.ceil(W / 2)
    ph, pw = max((Ho - 1) * 2 + 3 - H, 0), max((Wo - 1) * 2 + 3 - W, 0)
    xr = x.double().requires_grad_(True)
    ref = F.conv2d(F.pad(xr, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]), w.double(), stride=2)
    assert ref.shape[-2:] == (Ho, Wo)
    dy = torch.randn(N, O, Ho, Wo).bfloat16()
    ref.backward(dy.double())
    wt = w.permute(2, 3, 1, 0).contiguous()            # (3, 3, I, O): the transposed image maed_weight_std_fwd writes
    with patched():
        dx = ops.conv3x3_s2_dgrad(cl(dy), wt, H, W, ph // 2, pw // 2)
    assert dx.shape == (N, I, H, W) and torch.isfinite(dx.float()).all()
    assert (dx.double() - xr.grad).abs().max() <= 1.5e-2 * xr.grad.abs().max()


@pytest.mark.parametrize("N,I,O,H,W", [(4, 64, 64, 8, 8), (1, 64, 72, 16, 16), (4, 64, 64, 7, 7), (1, 128, 64, 15, 16), (2, 64, 136, 16, 8)])
def test_conv3x3_stride2_weight_gradient_through_gather_tables(N, I, O, H, W):
    """round 4: maed_conv3x3_s2_wgrad -- the TN weight-gradient kernel over input rows gathered through per-output-pixel tables (row of the top-left tap + 9-bit
    'tap inside the image' mask), against autograd through F.conv2d on the TF-SAME padded input; accumulates into the given fp32 slice."""
    import math
    torch.manual_seed(8)
    x = torch.randn(N, I, H, W).bfloat16()
    w = (torch.randn(O, I, 3, 3) * (9 * I) ** -0.5).bfloat16()
    Ho, Wo = math.ceil(H / 2), math.ceil(W / 2)
    assert (N * Ho * Wo) % 64 == 0
    ph, pw = max((Ho - 1) * 2 + 3 - H, 0), max((Wo - 1) * 2 + 3 - W, 0)
    wr = w.double().requires_grad_(True)
    ref = F.conv2d(F.pad(x.double(), [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]), wr, stride=2)
    dy = torch.randn(N, O, Ho, Wo).bfloat16()
    ref.backward(dy.double())
    dW0 = torch.randn(O, 3, 3, I)
    dW = dW0.clone()
    with patched():
        ops.conv3x3_s2_wgrad(cl(dy), cl(x), ph // 2, pw // 2, out=dW)
    got = (dW - dW0).permute(0, 3, 1, 2).double()
    assert (got - wr.grad).abs().max() <= 2e-3 * wr.grad.abs().max()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_weight_standardisation_tiled_kernel_64_filters_per_workgroup(dtype):
    """round 4: the transposed weight images by a tile transpose of the forward image (ws_transpose_kernel: 64 x 64 tiles through LDS, 128-byte runs) instead of
    element-wise strided stores.  Shapes: the stem's K = 147 (no transposed image), a 1x1 with two filter blocks, a 3x3 with a ragged filter block (O = 72) --
    forward image and transposed image against ATen; the one-kernel form (MAED_WS_TILED=0) must give the same bits."""
    torch.manual_seed(4)

    class Owner:
        _pending_backwards = 0
        grads_ready = None
        _gemm_convs = [1, 2]

        def __init__(self, ws):
            self._ws = ws
            self._w_std_t, self._dw_slices, self._dw_arena = {}, {}, None

        def conv_weights(self):
            return self._ws
    shapes = [(64, 3, 7, 7), (128, 64, 1, 1), (72, 64, 3, 3)]
    ws = [torch.randn(s) for s in shapes]
    res = {}
    for tiled in ("1", "0"):
        owner = Owner(ws)
        os.environ["MAED_WS_TILED"] = tiled
        try:
            with patched(), torch.no_grad():
                outs = ops.WeightStdFn.apply(owner, dtype, 1e-5, *ws)
                res[tiled] = ([o.clone() for o in outs], {i: t.clone() for i, t in owner._w_std_t.items()})       # (fp32 in the exact mode: no transposed images)
        finally:
            os.environ.pop("MAED_WS_TILED", None)
    tol = dict(rtol=1e-2, atol=1e-2) if dtype == torch.bfloat16 else dict(rtol=1e-5, atol=1e-5)
    for i, w in enumerate(ws):
        std, mean = torch.std_mean(w, dim=[1, 2, 3], keepdim=True, unbiased=False)
        ref = (w - mean) / (std + 1e-5)
        assert torch.allclose(res["1"][0][i].float(), ref, **tol), i
        assert torch.equal(res["1"][0][i], res["0"][0][i]), i
        if i in res["1"][1]:
            O, I, kh, kw = shapes[i]
            assert torch.allclose(res["1"][1][i].float(), ref.permute(2, 3, 1, 0).reshape(kh * kw * I, O), **tol), i
            assert torch.equal(res["1"][1][i], res["0"][1][i]), i
    assert (dtype == torch.bfloat16) == (sorted(res["1"][1]) == [1, 2])


def _stem_reference(x, w, dy=None):
    """fp64 F.conv2d on the TF-SAME padded frames (resnetv2.py:51-59, :74-93): 2 rows / columns before, 3 after for even extents"""
    wr = w.double().requires_grad_(True)
    ref = F.conv2d(F.pad(x.double(), [2, 3, 2, 3]), wr, stride=2)
    if dy is not None:
        ref.backward(dy.double())
    return ref.detach(), wr.grad


@pytest.mark.parametrize("N,H,W,stats", [(2, 32, 32, True), (1, 32, 64, False), (1, 48, 96, True), (1, 16, 256, True)])
def test_stem7x7s2_forward_from_padded_4slot_image(N, H, W, stats):
    """round 4: maed_stem_input(c_stride 4) + maed_stem7x7s2_fwd -- the stem convolution with the pixel operand loaded from global memory straight into the MFMA
    fragment layout (16 bytes = 2 pixels x 4 slots = the stride-2 step), against F.conv2d on the padded frames; the GroupNorm statistics of the rounded outputs
    (32 groups of 2 channels) from the epilogue.  2 / 4 / 3 tiles per wave."""
    torch.manual_seed(11)
    x = torch.randn(N, 3, H, W)
    w = (torch.randn(64, 3, 7, 7) * 147 ** -0.5).bfloat16()
    ref, _ = _stem_reference(x.bfloat16(), w)
    sums = torch.zeros(N, 32, 2, dtype=torch.float64) if stats else None
    with patched():
        assert ops.stem7x7s2_supported(H, W) and not ops.stem7x7s2_supported(H + 1, W) and not ops.stem7x7s2_supported(20, 20)
        xp = ops.stem_input(x, torch.bfloat16, 7, 2, own=True)
        assert xp.shape == (N, 4, H + 5, W + 6) and (xp[:, 3] == 0).all() and (xp[:, :, :, -1] == 0).all() and torch.equal(xp[:, :3, 2:H + 2, 2:W + 2], x.bfloat16())
        y = ops.StemConvFn.apply(xp, cl(w), None, sums, (H, W))
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert (y.double() - ref).abs().max() <= 1e-2 * ref.abs().max()
    if stats:
        yg = y.float().permute(0, 2, 3, 1).reshape(N, (H // 2) * (W // 2), 32, 2).double()
        want = torch.stack([yg.sum(dim=(1, 3)), (yg * yg).sum(dim=(1, 3))], dim=-1)
        assert torch.allclose(sums, want, rtol=1e-5, atol=1e-3), (sums - want).abs().max()


@pytest.mark.parametrize("N,H,W,wgs,slots", [(2, 32, 32, None, True), (1, 32, 64, None, False), (3, 48, 96, 5, True), (2, 32, 32, 3, False), (2, 32, 32, 3, True),
                                              (1, 16, 256, 3, True)])      # (last: cfg5's row width -- three DMA rounds per image, 86 KB of LDS)
def test_stem7x7s2_weight_gradient_by_lds_dma_and_transposing_reads(N, H, W, wgs, slots):
    """round 4: maed_stem7x7s2_wgrad -- one output row of dy and its seven input rows per work item, copied into LDS unchanged (LDS-DMA, swizzled dy chunks) and
    contracted over pixels through ds_read_b64_tr_b16 fragments; accumulates into the given fp32 slice.  One row per workgroup by default at these sizes; with
    MAED_OPT_STEM_WGRAD_WGS = 5 / 3 the double-buffered walk over 15 (14 for the last workgroup... 72 = 4 * 15 + 12) resp. 11 / 11 / 10 rows.  slots: per-workgroup partial
    results in scratch + the reduction pass (the product path) / atomics straight into the slice."""
    torch.manual_seed(12)
    x = torch.randn(N, 3, H, W)
    w = (torch.randn(64, 3, 7, 7) * 147 ** -0.5).bfloat16()
    dy = torch.randn(N, 64, H // 2, W // 2).bfloat16()
    _, gw = _stem_reference(x.bfloat16(), w, dy)
    dW0 = torch.randn(64, 147)
    dW = dW0.clone()
    with patched() as lib, option(lib, L.OPT_STEM_WGRAD_WGS, wgs or 512):
        xp = ops.stem_input(x, torch.bfloat16, 7, 2, own=True)
        dyc = cl(dy)
        sc = torch.full((lib.maed_stem7x7s2_wgrad_scratch_floats(N, H, W),), float("nan")) if slots else None
        assert sc is None or sc.numel() % (64 * 147) == 0 and sc.numel() > 0
        rc = lib.maed_stem7x7s2_wgrad(dyc.data_ptr(), xp.data_ptr(), dW.data_ptr(), sc.data_ptr() if sc is not None else None, N, H, W, L.BF16, None)
        assert rc == 0, lib.maed_last_error()
    got = (dW - dW0).view(64, 7, 7, 3).permute(0, 3, 1, 2).double()
    assert (got - gw).abs().max() <= 2e-3 * gw.abs().max(), (got - gw).abs().max() / gw.abs().max()


def test_stem7x7s2_autograd_node_fills_the_fp32_slice_and_rejects_other_geometries():
    torch.manual_seed(13)
    x = torch.randn(2, 3, 32, 32)
    w = (torch.randn(64, 3, 7, 7) * 147 ** -0.5).bfloat16()
    dy = torch.randn(2, 64, 16, 16).bfloat16()
    _, gw = _stem_reference(x.bfloat16(), w, dy)
    dw = torch.zeros(64, 147)
    with patched() as lib:
        xp = ops.stem_input(x, torch.bfloat16, 7, 2, own=True)
        ws = cl(w).requires_grad_(True)          # (autograd must see a differentiable input for the node's backward to run)
        y = ops.StemConvFn.apply(xp, ws, dw, None, (32, 32))
        y.backward(cl(dy))
        assert ws.grad is None                   # the gradient travels in the fp32 slice, not through autograd
        bad = torch.zeros(1, 20 + 5, 20 + 6, 4, dtype=torch.bfloat16)
        assert lib.maed_stem7x7s2_wgrad(dy.data_ptr(), bad.data_ptr(), dw.data_ptr(), None, 1, 20, 20, L.BF16, None) != 0
    got = dw.view(64, 7, 7, 3).permute(0, 3, 1, 2).double()
    assert (got - gw).abs().max() <= 2e-3 * gw.abs().max()


@pytest.mark.parametrize("N,H,W,wgs", [(2, 5, 8, None), (1, 3, 16, None), (3, 4, 24, 2), (2, 6, 56, 5), (1, 2, 64, 1)])
def test_conv3x3_weight_gradient_row_items_64_channels(N, H, W, wgs):
    """round 4: the 64 -> 64 channel weight gradient one image row at a time (conv3x3_rows.hip, taken by maed_conv3x3_wgrad for the stage-1 shape): rows in a
    ring of four LDS slots framed by zero pixels, wave = tap, transposing reads.  Image heights / widths that exercise the borders (H = 2, 3), k-steps with a zero
    tail (W = 8, 24, 56), the full 64-pixel row, workgroups that walk rows across frame boundaries (MAED_OPT_CONV3X3_ROWS_WGS), and accumulation into a non-zero
    slice; against autograd through F.conv2d.  MAED_OPT_CONV3X3_ROWS_WGS = 0 (the general TN kernel) gives the same numbers where it applies."""
    torch.manual_seed(21)
    x = torch.randn(N, 64, H, W).bfloat16()
    dy = torch.randn(N, 64, H, W).bfloat16()
    wr = (torch.randn(64, 64, 3, 3) * 576 ** -0.5).double().requires_grad_(True)
    F.conv2d(x.double(), wr, padding=1).backward(dy.double())
    dW0 = torch.randn(64, 3, 3, 64)
    dW = dW0.clone()
    with patched() as lib, option(lib, L.OPT_CONV3X3_ROWS_WGS, wgs or 256):
        ops.conv3x3_wgrad(cl(dy), cl(x), out=dW)
    got = (dW - dW0).permute(0, 3, 1, 2).double()
    assert (got - wr.grad).abs().max() <= 2e-3 * wr.grad.abs().max(), ((got - wr.grad).abs().max() / wr.grad.abs().max())


@pytest.mark.parametrize("NF,H,W,Cin,Cout", [(2, 14, 14, 64, 128), (1, 16, 16, 128, 136), (2, 11, 13, 64, 64)])
def test_conv3x3_one_frame_per_workgroup_is_bit_identical_to_the_128_row_tiles(NF, H, W, Cin, Cout):
    """round 6 (second session): conv3x3_frame_bf16_kernel -- one frame x 128 channels per workgroup, three-stage copy ring, eight waves of 32 pixel rows (the last
    ones partly or wholly past the frame), for feature maps of 129 .. 256 pixels -- against the 128 x 128 tiles (MAED_OPT_CONV3X3_FRAME = 0): same K order, same
    MFMA shape, so the SAME bits; forward with GroupNorm statistics, with an `add` operand, and as the input gradient (transposed weight image, flipped taps);
    a ragged last column tile (Cout = 136) and a 64-column one"""
    torch.manual_seed(7)
    x = torch.randn(NF, Cin, H, W).bfloat16().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(Cout, 3, 3, Cin) * (9 * Cin) ** -0.5).bfloat16()
    dy = torch.randn(NF, Cout, H, W).bfloat16().contiguous(memory_format=torch.channels_last)
    wimg = wt.permute(1, 2, 3, 0).contiguous()             # (3, 3, Cin, Cout): the transposed image maed_weight_std_fwd writes beside the forward weight
    addt = torch.randn(NF, Cout, H, W).bfloat16().contiguous(memory_format=torch.channels_last)
    cpg = Cout // 32
    gn_ok = Cout % 32 == 0 and cpg >= 2 and (cpg & (cpg - 1)) == 0
    out = {}
    with patched() as lib:
        for mode in (0, 2):
            with option(lib, L.OPT_CONV3X3_FRAME, mode):
                sums = torch.zeros(NF, 32, 2, dtype=torch.float64)
                y = ops.conv3x3(x, wt, 1, gn_sums=sums if gn_ok else None)
                ya = ops.conv3x3(x, wt, 1, add=addt)
                dx = ops.conv3x3(dy, wimg, 1, w_layout=1) if Cout % 64 == 0 else y
                out[mode] = (y, ya, dx, sums)
    for a, b in zip(out[0][:3], out[2][:3]):
        assert a.shape == b.shape and torch.equal(a, b)
    assert torch.allclose(out[0][3], out[2][3], rtol=1e-6, atol=1e-4)       # (fp32 partial sums per lane over other row sets: the statistics agree to fp32 rounding)
    ref = F.conv2d(x.double(), wt.permute(0, 3, 1, 2).double(), padding=1)
    assert torch.allclose(out[2][0].double(), ref, rtol=2e-2, atol=2e-2)
    if Cout % 64 == 0:
        assert out[2][2].shape == (NF, Cin, H, W)
