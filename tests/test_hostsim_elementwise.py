"""STE element-wise kernels (maed_amd/csrc/elementwise.hip: attentive-addition mean / mix forward+backward, token embedding
forward+backward, transpose-cast) on the host simulator against the CPU oracle -- the same comparisons tests/test_gpu_kernels.py
runs on the real library, in fp32."""
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
