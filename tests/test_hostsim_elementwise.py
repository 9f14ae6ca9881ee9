"""STE element-wise kernels (maed_amd/csrc/elementwise.hip: attentive-addition mean / mix forward+backward, token embedding
forward+backward, transpose-cast) on the host simulator against the CPU oracle -- the same comparisons tests/test_gpu_kernels.py
runs on the real library, in fp32."""
import pytest
import torch

from oracle import maed_ref as R
from maed_amd import ops

from _hostsim import patched


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def test_st_mix_forward_backward_on_simulator():
    Fr, P, C = 3, 11, 128
    xs, xt, logits = rnd(Fr, P, C, seed=1), rnd(Fr, P, C, seed=2), rnd(Fr, 2 * C, seed=3)
    a, b, lg = xs.double().requires_grad_(True), xt.double().requires_grad_(True), logits.double().requires_grad_(True)
    alpha = lg.reshape(Fr, 1, C, 2).softmax(-1)                  # vision_transformer.py:154-158
    ref = b * alpha[:, :, :, 1] + a * alpha[:, :, :, 0]
    dmix = rnd(Fr, P, C, seed=4)
    ref.backward(dmix.double())
    W = rnd(2 * C, 2 * C, seed=5, scale=0.05)
    dmeans_ref = lg.grad @ W.double()
    with patched():
        means = ops.st_colmean(xs, xt)
        mix = ops.st_mix_fwd(xs, xt, logits)
        dxs, dxt, dlog = ops.st_mix_bwd(dmix, xs, xt, logits, lambda dl: dl.float() @ W)
    assert torch.allclose(means.double(), torch.cat([xs, xt], -1).double().mean(1), atol=1e-6)
    assert torch.allclose(mix.double(), ref.detach(), atol=1e-5)
    assert torch.allclose(dlog.double(), lg.grad, atol=1e-4)
    assert torch.allclose(dxs.double(), a.grad + dmeans_ref[:, None, :C] / P, atol=1e-5)
    assert torch.allclose(dxt.double(), b.grad + dmeans_ref[:, None, C:] / P, atol=1e-5)


def test_embed_add_forward_backward_on_simulator():
    N, T, P, C = 2, 3, 5, 128
    patch = rnd(N * T, P - 1, C, seed=1)
    prm = {"cls_token": rnd(1, 1, C, seed=2), "pos_embed": rnd(1, P, C, seed=3), "temp_embed": rnd(1, 16, 1, C, seed=4)}
    leaves = {k: v.double().requires_grad_(True) for k, v in prm.items()}
    pt = patch.double().requires_grad_(True)
    ref = R.embed_tokens(pt, leaves, "", T)                      # vision_transformer.py:392-399
    dtok = rnd(N * T, P, C, seed=5)
    ref.backward(dtok.double())
    args = [t.clone().requires_grad_(True) for t in (patch, prm["cls_token"], prm["pos_embed"], prm["temp_embed"])]
    with patched():
        tok = ops.EmbedAddFn.apply(*args, T)
        tok.backward(dtok)
    assert torch.allclose(tok.double(), ref.detach(), atol=1e-6)
    for got, want in zip(args, (pt, leaves["cls_token"], leaves["pos_embed"], leaves["temp_embed"])):
        assert torch.allclose(got.grad.double(), want.grad, atol=1e-5)


def test_transpose_cast_on_simulator():
    x = rnd(70, 45, seed=6)
    colsum = torch.zeros(45)
    with patched():
        xt, xc = ops.transpose_cast(x, torch.float32, want_t=True, want_c=True, colsum=colsum, pad_to=64)
    assert xt.shape == (45, 128) and torch.equal(xt[:, :70], x.t()) and torch.equal(xt[:, 70:], torch.zeros(45, 58))
    assert torch.equal(xc, x) and torch.allclose(colsum, x.sum(0), atol=1e-5)


@pytest.mark.parametrize("F_,P,C", [(3, 197, 256), (2, 257, 384), (2, 20, 128)])
def test_attentive_addition_fused_one_launch_matches_the_four_launch_sequence(F_, P, C):
    """round 4: maed_st_fused_fwd / _bwd (token means + ts_attn Linear + pair softmax + mix in one launch; backward likewise) against the separate entry points
    they replace in the fused STE block -- same bf16 roundings (means, dlogits, d(means)), so the results agree to summation order: C / 128 = 2, 3 workgroups per
    frame exchange through `ex` (the simulator runs the two halves of the kernel as two launches), P = 257 takes the 9-chunk instantiation."""
    from maed_amd import _lib as L
    g = torch.Generator().manual_seed(3)
    bf = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).bfloat16()
    xs, xt, dmix = bf(F_, P, C), bf(F_, P, C), bf(F_, P, C)
    w, b = bf(2 * C, 2 * C, sc=(2 * C) ** -0.5), torch.randn(2 * C, generator=g) * 0.1
    wt = w.t().contiguous()
    p = lambda t: ops._p(t)          # (resolved at call time: patched() swaps the pointer helper)
    with patched() as lib:
        assert lib.maed_st_fused_supported(P, C, L.BF16) == 1
        # the four-launch sequence
        means0 = ops.st_colmean(xs, xt)
        logits0 = ops.gemm_nt(means0, w, L.EPI_STORE_F32, bias=b)
        mix0 = ops.st_mix_fwd(xs, xt, logits0)
        dmeans_fn = lambda dlog: ops.gemm_nt(dlog, wt, L.EPI_STORE)
        dxs0, dxt0, dlog0 = ops.st_mix_bwd(dmix, xs, xt, logits0, dmeans_fn)
        # one launch per direction
        means1, logits1, mix1 = torch.empty(F_, 2 * C, dtype=torch.bfloat16), torch.empty(F_, 2 * C), torch.empty_like(xs)
        sync, ex = torch.full((F_ * 16,), 7, dtype=torch.int32), torch.empty(F_ * 2 * C)
        L.check(lib.maed_st_fused_fwd(p(xs), p(xt), p(w), p(b), p(means1), p(logits1), p(mix1), p(sync), p(ex), F_, P, C, L.BF16, None), "st_fused_fwd")
        dlog1, dxs1, dxt1 = torch.empty(F_, 2 * C, dtype=torch.bfloat16), torch.empty_like(xs), torch.empty_like(xs)
        L.check(lib.maed_st_fused_bwd(p(dmix), p(xs), p(xt), p(logits1), p(wt), p(dlog1), p(dxs1), p(dxt1), p(sync), p(ex), F_, P, C, L.BF16, None), "st_fused_bwd")
    f = lambda t: t.float()
    assert (f(means1) - f(means0)).abs().max() <= 2.0 ** -7 * f(means0).abs().max()            # at most one bf16 ulp (summation order before the rounding)
    assert torch.allclose(logits1, logits0, rtol=0, atol=2e-3 * logits0.abs().max().item())
    assert torch.allclose(f(mix1), f(mix0), rtol=0, atol=2e-2 * f(mix0).abs().max().item())
    assert torch.allclose(f(dlog1), f(dlog0), rtol=0, atol=2e-2 * f(dlog0).abs().max().item())
    assert torch.allclose(f(dxs1), f(dxs0), rtol=0, atol=2e-2 * f(dxs0).abs().max().item()) and torch.allclose(f(dxt1), f(dxt0), rtol=0, atol=2e-2 * f(dxt0).abs().max().item())
    # and against fp64 on the same inputs
    xs64, xt64 = xs.double(), xt.double()
    m64 = torch.cat([xs64.mean(1), xt64.mean(1)], 1)
    lg64 = m64 @ w.double().t() + b.double()
    a0 = torch.softmax(lg64.view(F_, C, 2), -1)[..., 0][:, None, :]
    mix64 = xs64 * a0 + xt64 * (1 - a0)
    assert (f(mix1).double() - mix64).abs().max() <= 2e-2 * mix64.abs().max()
