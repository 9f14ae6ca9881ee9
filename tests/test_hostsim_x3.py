"""The split-bf16 ("bf16x3" / "bf16x6") fp32 GEMM family of csrc/gemm_x3.hip on the host simulator: fp32 operands split into 2 / 3 bf16 planes
while staging, 3 / 6 MFMAs per product.  Checks the plane / fragment / staging index algebra of the NT kernel (all fused epilogues, ragged
edges, split-K atomics, narrow tiles, GroupNorm statistics), the gathered 3x3 implicit GEMM (forward, transposed-image input gradient, any
stride), the transposing TN weight-gradient kernel (+ bias gradient, ragged M, the 3x3 tap mask) -- and that the ERROR is what the scheme
promises: ~2^-16 of |a||b| for x3, fp32 level for x6 (vs 2^-8 for plain bf16), measured against fp64."""
import pytest
import torch
import torch.nn.functional as F

from maed_amd import _lib as L
from maed_amd import ops

from _hostsim import patched


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def gelu(x):
    return 0.5 * x * (1 + torch.erf(x / 2 ** 0.5))


def dgelu(x):
    return 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5


def bound(A, B, np_):
    """elementwise error bound of the split product: eps * (|A| |B|^T) with eps = 2^-15 (x3: 3 * 2^-18 dropped terms + fp32 accumulation) / 2^-21 (x6)"""
    return (A.abs().double() @ B.abs().double().t()) * (2.0 ** -15 if np_ == 2 else 2.0 ** -21)


@pytest.fixture
def x3_mode():
    """process-wide fp32 matmul mode = bf16x3 for the duration of a test"""
    old = ops.get_float32_matmul_precision()
    ops.set_float32_matmul_precision("bf16x3")
    yield
    ops.set_float32_matmul_precision(old)


@pytest.mark.parametrize("impl,np_", [(L.IMPL_X3, 2), (L.IMPL_X6, 3)])
@pytest.mark.parametrize("M,N,K", [(130, 136, 96), (64, 256, 32), (257, 72, 160)])
def test_gemm_nt_x3_store_and_error_level(impl, np_, M, N, K):
    A, B, bias = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    ref = A.double() @ B.double().t() + bias.double()
    with patched():
        out = ops.gemm_nt(A, B, L.EPI_STORE, bias=bias, impl=impl)
    err = (out.double() - ref).abs()
    assert (err <= bound(A, B, np_) + 1e-6 * ref.abs()).all(), (err.max().item(), bound(A, B, np_).max().item())
    # and it is NOT merely bf16: plain bf16 operands would be ~2^-9 relative
    assert err.max() <= 1e-4 * ref.abs().max()


@pytest.mark.parametrize("epi", ["gelu", "resid", "dgelu", "store_f32", "tanh", "add", "add_mask"])
def test_gemm_nt_x3_fused_epilogues(epi):
    M, N, K = 96, 200, 64
    A, B, bias = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=K ** -0.5), rnd(N, seed=6)
    acc = (A.double() @ B.double().t()).float()
    tol = dict(rtol=1e-4, atol=1e-4)
    with patched():
        if epi == "gelu":
            out, pre = ops.gemm_nt(A, B, L.EPI_GELU, bias=bias, impl=L.IMPL_X3)
            assert torch.allclose(pre, acc + bias, **tol) and torch.allclose(out, gelu(pre), **tol)
        elif epi == "resid":
            aux = rnd(M, N, seed=7)
            assert torch.allclose(ops.gemm_nt(A, B, L.EPI_RESID_F32, bias=bias, aux=aux, impl=L.IMPL_X3), aux + acc + bias, **tol)
        elif epi == "dgelu":
            aux = rnd(M, N, seed=8)
            assert torch.allclose(ops.gemm_nt(A, B, L.EPI_MUL_DGELU, aux=aux, impl=L.IMPL_X3), acc * dgelu(aux), **tol)
        elif epi == "store_f32":
            assert torch.allclose(ops.gemm_nt(A, B, L.EPI_STORE_F32, bias=bias, impl=L.IMPL_X3), acc + bias, **tol)
        elif epi == "tanh":
            assert torch.allclose(ops.gemm_nt(A, B, L.EPI_TANH, bias=bias, impl=L.IMPL_X3), torch.tanh(acc + bias), **tol)
        elif epi == "add":
            aux = rnd(M, N, seed=9)
            assert torch.allclose(ops.gemm_nt(A, B, L.EPI_ADD, aux=aux, impl=L.IMPL_X3), acc + aux, **tol)
        else:
            aux = rnd(M, N, seed=9)
            keep = torch.rand(M, N, generator=torch.Generator().manual_seed(34)) > 0.4
            bits = (keep.view(M, N // 8, 8).to(torch.uint8) << torch.arange(8, dtype=torch.uint8)).sum(-1).to(torch.uint8).contiguous()
            assert torch.allclose(ops.gemm_nt(A, B, L.EPI_ADD, aux=aux, out2=bits, impl=L.IMPL_X3), acc + aux * keep, **tol)


def test_gemm_nt_x3_splitk_atomic_and_auto_dispatch(x3_mode):
    M, N, K = 70, 128, 256
    A, B = rnd(M, K, seed=10), rnd(N, K, seed=11, scale=K ** -0.5)
    ref = A.double() @ B.double().t()
    with patched():
        out = ops.gemm_nt(A, B, L.EPI_ATOMIC_F32, splitk=2)                  # AUTO + mode bf16x3 -> the split kernel, K range per z
        exact = ops.gemm_nt(A, B, L.EPI_STORE, impl=L.IMPL_VALU)            # the exact kernel stays selectable
        few = ops.gemm_nt(A, B, L.EPI_STORE)                                # few tiles, K >= 256: bias fill + K slices on the split kernel (atomic epilogue)
        fewb = ops.gemm_nt(A, B, L.EPI_STORE, bias=rnd(N, seed=14))
        Ab, Bb = rnd(1024, 64, seed=12), rnd(768, 64, seed=13)
        many = ops.gemm_nt(Ab, Bb, L.EPI_STORE)                             # 48 tiles: AUTO takes the split kernel
    assert (out.double() - ref).abs().max() <= 1e-4 * ref.abs().max()
    assert torch.allclose(exact.double(), ref, rtol=1e-5, atol=1e-5)
    errf = (few.double() - ref).abs().max()
    assert 1e-8 * ref.abs().max() < errf <= 1e-4 * ref.abs().max(), errf      # split arithmetic
    assert (fewb.double() - ref - rnd(N, seed=14).double()).abs().max() <= 1e-4 * ref.abs().max()
    refm = Ab.double() @ Bb.double().t()
    errm = (many.double() - refm).abs().max()
    assert 1e-7 * refm.abs().max() < errm <= 1e-4 * refm.abs().max()          # split arithmetic, not the exact kernel and not bf16


@pytest.mark.parametrize("Cout,hw", [(64, 128), (256, 196)])
def test_conv1x1_x3_with_groupnorm_statistics(x3_mode, Cout, hw):
    Fr, Cin = 2, 64
    M = Fr * hw
    x, w = rnd(M, Cin, seed=20), rnd(Cout, Cin, seed=21, scale=Cin ** -0.5)
    sums = torch.zeros(Fr, 32, 2, dtype=torch.float64)
    y = torch.empty(M, Cout)
    with patched() as lib:
        L.check(lib.maed_conv1x1_fwd(x.data_ptr(), Cin, w.data_ptr(), Cin, M, Cout, Cin, y.data_ptr(), Cout, hw, sums.data_ptr(), L.F32, None), "conv1x1")
    ref = x.double() @ w.double().t()
    assert (y.double() - ref).abs().max() <= 1e-4 * ref.abs().max()
    g = y.double().view(Fr, hw, 32, Cout // 32)
    want = torch.stack([g.sum((1, 3)), (g * g).sum((1, 3))], -1)
    assert torch.allclose(sums, want, rtol=1e-5, atol=1e-3)          # per-lane fp32 partials, fp64 across lanes


def _conv_ref(x, w, stride):
    """TF-SAME 3x3 convolution in fp64 (resnetv2.py:51-59 padding)"""
    H, W = x.shape[-2:]
    Ho, Wo = -(-H // stride), -(-W // stride)
    ph, pw = max((Ho - 1) * stride + 3 - H, 0), max((Wo - 1) * stride + 3 - W, 0)
    xp = F.pad(x.double(), [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])
    return F.conv2d(xp, w.double(), None, stride)


@pytest.mark.parametrize("stride,H,W,Cin,Cout", [(1, 9, 7, 32, 64), (2, 10, 9, 64, 136), (1, 6, 6, 64, 72)])
def test_conv3x3_x3_forward(x3_mode, stride, H, W, Cin, Cout):
    N = 3
    x = rnd(N, Cin, H, W, seed=30).contiguous(memory_format=torch.channels_last)
    w = rnd(Cout, Cin, 3, 3, seed=31, scale=(9 * Cin) ** -0.5)
    w_taps = w.permute(0, 2, 3, 1).contiguous()
    ref = _conv_ref(x, w, stride)
    with patched():
        y = ops.conv3x3(x, w_taps, stride)
        add = rnd(*ref.shape, seed=32).contiguous(memory_format=torch.channels_last)
        y2 = ops.conv3x3(x, w_taps, stride, add=add)
    assert y.shape == ref.shape and (y.double() - ref).abs().max() <= 1e-4 * ref.abs().max()
    assert (y2.double() - ref - add.double()).abs().max() <= 1e-4 * ref.abs().max()


def test_conv3x3_x3_input_gradient_from_transposed_image_and_wgrad(x3_mode):
    N, Cin, Cout, H, W = 2, 32, 64, 8, 8          # N*H*W = 128 (the weight-gradient kernel needs a multiple of 32)
    x = rnd(N, Cin, H, W, seed=40).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = rnd(Cout, Cin, 3, 3, seed=41, scale=(9 * Cin) ** -0.5).requires_grad_(True)
    dy = rnd(N, Cout, H, W, seed=42).contiguous(memory_format=torch.channels_last)
    y = F.conv2d(x.double(), w.double(), None, 1, 1)
    gx, gw = torch.autograd.grad(y, (x, w), dy.double())
    wt = w.detach().permute(2, 3, 1, 0).contiguous()          # transposed image (3,3,I,O) as maed_weight_std_fwd writes it
    with patched():
        dx = ops.conv3x3(dy, wt, 1, w_layout=1)
        dW = ops.conv3x3_wgrad(dy, x.detach())
    assert (dx.double() - gx).abs().max() <= 1e-4 * gx.abs().max()
    assert (dW.permute(0, 3, 1, 2).double() - gw).abs().max() <= 1e-4 * gw.abs().max()


@pytest.mark.parametrize("M,N,K", [(200, 136, 72), (64, 128, 128), (37, 8, 8), (130, 64, 256)])
def test_gemm_tn_x3_wgrad_and_bias(x3_mode, M, N, K):
    Y, X = rnd(M, N, seed=12), rnd(M, K, seed=13)
    dW0, db0 = rnd(N, K, seed=14), rnd(N, seed=15)
    dW, db = dW0.clone(), db0.clone()
    with patched():
        ops.gemm_tn_wgrad(Y, X, dW=dW, dbias=db)          # ACCUMULATES into dW / dbias
    ref = Y.double().t() @ X.double()
    assert ((dW - dW0).double() - ref).abs().max() <= 1e-4 * ref.abs().max()
    assert torch.allclose(db, db0 + Y.sum(0), rtol=1e-5, atol=1e-4)      # the bias gradient is exact fp32


def test_gemm_tn_x6_is_fp32_level():
    M, N, K = 256, 128, 128
    Y, X = rnd(M, N, seed=16), rnd(M, K, seed=17)
    ref = Y.double().t() @ X.double()
    old = ops.get_float32_matmul_precision()
    try:
        ops.set_float32_matmul_precision("bf16x6")
        with patched():
            dW = ops.gemm_tn_wgrad(Y, X)
    finally:
        ops.set_float32_matmul_precision(old)
    assert (dW.double() - ref).abs().max() <= 2e-6 * ref.abs().max()


def test_gemm_tn_x3_rejects_operands_past_4gb(x3_mode):
    """the split TN kernel addresses rows with 32-bit byte offsets against a 4 GB buffer extent -- an operand past that must be refused (as the NT /
    convolution launchers do), not read wrapped.  A row stride of 2^26 floats makes 64 rows span 16 GB without allocating them: the check fires before any access."""
    Y, X, dW = rnd(64, 8, seed=1), rnd(64, 8, seed=2), torch.zeros(8, 8)
    with patched():
        rc = L.lib().maed_gemm_tn_wgrad(ops._p(Y), 1 << 26, ops._p(X), 8, 64, 8, 8, ops._p(dW), 8, None, ops.mm_code(torch.float32), None)
        assert rc != 0 and b"4 GB" in L.lib().maed_last_error()
        rc = L.lib().maed_gemm_tn_wgrad(ops._p(Y), 8, ops._p(X), 1 << 26, 64, 8, 8, ops._p(dW), 8, None, ops.mm_code(torch.float32), None)
        assert rc != 0 and b"4 GB" in L.lib().maed_last_error()
    assert dW.abs().max() == 0


def test_bf16x1_backward_engine_one_plane_products(x3_mode):
    """round 4, the mixed mode "bf16x3 forward / bf16 backward": MAED_F32X1 = fp32 operands, ONE bf16 plane, one MFMA per product -- the same kernels with NP = 1.
    The error is what a bf16 product has (2^-8 of |a||b|), not the split product's 2^-16: checked from both sides, for the NT GEMM (input gradients, incl. the
    masked-add epilogue), the TN weight gradient and the 3x3 input gradient -- and ops.bwd_prec routes the backward there when the option is set."""
    A, B = rnd(200, 96, seed=21), rnd(136, 96, seed=22)
    ref = A.double() @ B.double().t()
    with patched():
        o1 = ops.gemm_nt(A, B, prec="bf16x1")
        o3 = ops.gemm_nt(A, B, prec="bf16x3")
    bnd = A.abs().double() @ B.abs().double().t()
    e1, e3 = ((o1.double() - ref).abs() / bnd).max().item(), ((o3.double() - ref).abs() / bnd).max().item()
    assert e3 < 2.0 ** -14 < e1 < 2.0 ** -7, (e1, e3)
    assert torch.allclose(o1.double(), A.bfloat16().double() @ B.bfloat16().double().t(), rtol=0, atol=1e-4 * ref.abs().max())     # = the product of the bf16-rounded operands
    Y, X = rnd(200, 72, seed=23), rnd(200, 136, seed=24)
    with patched():
        dW = ops.gemm_tn_wgrad(Y, X, prec="bf16x1")
    assert torch.allclose(dW.double(), Y.bfloat16().double().t() @ X.bfloat16().double(), rtol=0, atol=1e-4 * dW.abs().max().item())
    x = rnd(1, 64, 7, 6, seed=25).contiguous(memory_format=torch.channels_last)
    w = rnd(64, 3, 3, 64, seed=26, scale=0.05)
    with patched():
        y1 = ops.conv3x3(x, w, 1, prec="bf16x1")
    yr = F.conv2d(x.bfloat16().double(), w.permute(0, 3, 1, 2).bfloat16().double(), padding=1)
    assert torch.allclose(y1.double(), yr, rtol=0, atol=1e-4 * yr.abs().max().item())
    assert ops.bwd_prec(None) is None and ops.bwd_prec("bf16x6") == "bf16x3"
    ops.set_float32_backward_precision("bf16x1")
    try:
        assert ops.bwd_prec(None) == "bf16x1" and ops.bwd_prec("bf16x6") == "bf16x1" and ops.get_float32_backward_precision() == "bf16x1"
    finally:
        ops.set_float32_backward_precision(None)


def test_f32_library_convolutions_need_the_split_mode():
    """in the exact mode the entry points without an exact fp32 kernel refuse fp32 (loudly: no silent precision change)"""
    assert ops.get_float32_matmul_precision() == "exact"
    Y, X = rnd(64, 8, seed=1), rnd(64, 8, seed=2)
    with patched():
        with pytest.raises(L.MaedHipError, match="split-bf16"):
            ops.gemm_tn_wgrad(Y, X)


# ---- attention: fp32 q/k/v, split-bf16 contractions (csrc/attn_x3.hip) ------------------------------------------------------------------
@pytest.mark.parametrize("impl,tol", [(L.IMPL_X3, 2e-4), (L.IMPL_X6, 5e-6)])
@pytest.mark.parametrize("Fr,L_,H", [(2, 5, 2), (1, 197, 2), (1, 64, 1), (1, 130, 1)])
def test_attn_x3_fwd_bwd_vs_fp64_oracle(impl, tol, Fr, L_, H):
    """the K/V-tiled kernels on fp32 operands: one partial tile; cfg3's 197 tokens (2 row tiles x 4 streamed tiles, ragged ends); exactly one full
    tile; a third wave with 2 rows.  Forward, log-sum-exp, dq / dk / dv and the accumulate flag against fp64 autograd through the oracle."""
    from oracle import maed_ref as R
    qkv = rnd(Fr, L_, 3 * 64 * H, seed=L_)
    do = rnd(Fr, L_, 64 * H, seed=4)
    x = qkv.double().requires_grad_(True)
    qq, kk, vv = R.split_qkv(x, H)
    oref = R.attention_spatial(qq, kk, vv, 64 ** -0.5)
    lse_ref = torch.logsumexp((qq @ kk.transpose(-2, -1)) * 64 ** -0.5, dim=-1)
    oref.backward(do.double())
    with patched():
        o, lse = ops.attn_spatial_fwd(qkv, H, impl)
        dqkv = ops.attn_spatial_bwd(qkv, o, do, lse, H, impl=impl)
        acc = ops.attn_spatial_bwd(qkv, o, do, lse, H, dqkv=dqkv.clone(), accumulate=True, impl=impl)
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
    assert rel(o, oref.detach()) <= tol, rel(o, oref.detach())
    assert (lse.double() - lse_ref.detach()).abs().max() <= 10 * tol
    assert rel(dqkv, x.grad) <= 2 * tol, rel(dqkv, x.grad)
    assert rel(acc, 2 * x.grad) <= 2 * tol


@pytest.mark.parametrize("N,T,P,H", [(1, 16, 9, 1), (2, 8, 5, 2), (1, 32, 3, 1), (2, 16, 2, 2)])
def test_attn_temporal_x3_fwd_bwd_vs_fp64_oracle(N, T, P, H):
    """temporal attention on one-tile virtual sequences with split-bf16 contractions (attn_tm_x3_fwd / _bwd): two tokens x 16 frames per tile with an odd
    token count (the last tile holds one token: absent rows are masked), four tokens x 8 frames, one token x 32 frames; forward, log-sum-exp, dq / dk / dv
    and the accumulate flag against fp64 autograd through the oracle -- per call ("bf16x3" dtype code) and through the process-wide mode"""
    from oracle import maed_ref as R
    Fr = N * T
    qkv = rnd(Fr, P, 3 * 64 * H, seed=T + P)
    do = rnd(Fr, P, 64 * H, seed=4)
    x = qkv.double().requires_grad_(True)
    qq, kk, vv = R.split_qkv(x, H)
    oref = R.attention_temporal(qq, kk, vv, T, 64 ** -0.5)
    oref.backward(do.double())
    with patched():
        o, lse = ops.attn_temporal_fwd(qkv, H, T, prec="bf16x3")
        dqkv = ops.attn_temporal_bwd(qkv, o, do, lse, H, T, prec="bf16x3")
        acc = ops.attn_temporal_bwd(qkv, o, do, lse, H, T, dqkv=dqkv.clone(), accumulate=True, prec="bf16x3")
        o_exact, lse_exact = ops.attn_temporal_fwd(qkv, H, T)                  # process-wide mode = exact: the VALU kernels
        old = ops.get_float32_matmul_precision()
        try:
            ops.set_float32_matmul_precision("bf16x3")
            o_mode, _ = ops.attn_temporal_fwd(qkv, H, T)
        finally:
            ops.set_float32_matmul_precision(old)
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
    tol = 2e-4
    assert rel(o, oref.detach()) <= tol, rel(o, oref.detach())
    assert 1e-7 < rel(o, o_exact), "the split kernel did not run (bit-identical to the exact kernel)"
    assert torch.equal(o_mode, o)
    assert (lse - lse_exact).abs().max() <= 10 * tol
    assert rel(dqkv, x.grad) <= 2 * tol, rel(dqkv, x.grad)
    assert rel(acc, 2 * x.grad) <= 2 * tol


def test_attn_temporal_x3_other_frame_counts_and_bf16x6_keep_the_exact_kernels():
    """T = 3 does not tile into 32 rows and bf16x6 has no temporal split kernel: both run the exact fp32 kernels (never less accurate than asked for)"""
    qkv = rnd(6, 5, 3 * 64, seed=1)
    with patched():
        ref, _ = ops.attn_temporal_fwd(qkv, 1, 3)
        a, _ = ops.attn_temporal_fwd(qkv, 1, 3, prec="bf16x3")
        b, _ = ops.attn_temporal_fwd(qkv[:4].contiguous(), 1, 2, prec="bf16x6")
        bref, _ = ops.attn_temporal_fwd(qkv[:4].contiguous(), 1, 2)
    assert torch.equal(a, ref) and torch.equal(b, bref)

