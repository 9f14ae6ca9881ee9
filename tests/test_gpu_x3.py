"""GPU parity tests of the split-bf16 fp32 matmul family (csrc/gemm_x3.hip; MAED_OPT_F32_MATMUL = bf16x3 / bf16x6): the fp32 operands' matrix
products on the bf16 matrix cores, against fp64 on the same inputs.  Tolerances are the scheme's promise, relative to the largest output:
1e-4 for bf16x3 (error ~2^-16 per product; plain bf16 would be ~4e-3), 2e-6 for bf16x6 (fp32 level)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from _util import DEV, note, report, rnd  # noqa: E402

MODES = [("bf16x3", 1e-4), ("bf16x6", 2e-6)]


@pytest.fixture(params=MODES, ids=[m for m, _ in MODES])
def mode(request):
    from maed_amd import ops
    old = ops.get_float32_matmul_precision()
    ops.set_float32_matmul_precision(request.param[0])
    yield request.param
    ops.set_float32_matmul_precision(old)


def gelu(x):
    return 0.5 * x * (1 + torch.erf(x / 2 ** 0.5))


def dgelu(x):
    return 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5


@pytest.mark.parametrize("M,N,K", [(25216, 1536, 512), (3000, 520, 2048), (197 * 5, 64, 64), (6272, 1024, 256)])
def test_gemm_nt_x3_vs_fp64(mode, M, N, K):
    from maed_amd import ops, _lib as L
    name, tol = mode
    A, B, bias = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    ref = A.double() @ B.double().t() + bias.double()
    out = ops.gemm_nt(A.to(DEV), B.to(DEV), L.EPI_STORE, bias=bias.to(DEV))          # AUTO follows the process-wide mode
    report(f"gemm_nt {name} [{M}x{N}x{K}] vs fp64", out, ref, rtol=0, atol=tol * ref.abs().max().item())
    exact = ops.gemm_nt(A.to(DEV), B.to(DEV), L.EPI_STORE, bias=bias.to(DEV), impl=L.IMPL_VALU)
    e_split, e_exact = (out.double().cpu() - ref).abs().max().item(), (exact.double().cpu() - ref).abs().max().item()
    note(f"gemm_nt {name} [{M}x{N}x{K}]: max err split {e_split:.3e}, exact-f32 VALU kernel {e_exact:.3e}")


@pytest.mark.parametrize("epi", ["gelu", "resid", "dgelu", "store_f32", "tanh", "add", "atomic"])
def test_gemm_nt_x3_epilogues(mode, epi):
    from maed_amd import ops, _lib as L
    name, tol = mode
    M, N, K = 197 * 16 + 5, 1032, 512
    A, B, bias = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=K ** -0.5), rnd(N, seed=6)
    acc = A.double() @ B.double().t()
    a, b, bs = A.to(DEV), B.to(DEV), bias.to(DEV)
    at = tol * acc.abs().max().item() * 4
    if epi == "gelu":
        out, pre = ops.gemm_nt(a, b, L.EPI_GELU, bias=bs)
        report(f"x3 gelu.pre {name}", pre, acc + bias.double(), rtol=0, atol=at)
        report(f"x3 gelu.act {name}", out, gelu(pre.double().cpu()), rtol=1e-5, atol=1e-5)
    elif epi == "resid":
        aux = rnd(M, N, seed=7)
        report(f"x3 resid {name}", ops.gemm_nt(a, b, L.EPI_RESID_F32, bias=bs, aux=aux.to(DEV)), aux.double() + acc + bias.double(), rtol=0, atol=at)
    elif epi == "dgelu":
        aux = rnd(M, N, seed=8)
        report(f"x3 dgelu {name}", ops.gemm_nt(a, b, L.EPI_MUL_DGELU, aux=aux.to(DEV)), acc * dgelu(aux.double()), rtol=1e-5, atol=at)
    elif epi == "store_f32":
        report(f"x3 store_f32 {name}", ops.gemm_nt(a, b, L.EPI_STORE_F32, bias=bs), acc + bias.double(), rtol=0, atol=at)
    elif epi == "tanh":
        report(f"x3 tanh {name}", ops.gemm_nt(a, b, L.EPI_TANH, bias=bs), torch.tanh(acc + bias.double()), rtol=0, atol=max(at, 1e-6))
    elif epi == "add":
        aux = rnd(M, N, seed=9)
        keep = torch.rand(M, N, generator=torch.Generator().manual_seed(34)) > 0.4
        bits = (keep.view(M, N // 8, 8).to(torch.uint8) << torch.arange(8, dtype=torch.uint8)).sum(-1).to(torch.uint8).contiguous()
        report(f"x3 add {name}", ops.gemm_nt(a, b, L.EPI_ADD, aux=aux.to(DEV)), acc + aux.double(), rtol=0, atol=at)
        report(f"x3 add+mask {name}", ops.gemm_nt(a, b, L.EPI_ADD, aux=aux.to(DEV), out2=bits.to(DEV)), acc + (aux * keep).double(), rtol=0, atol=at)
    else:
        out = torch.zeros(M, N, device=DEV)
        ops.gemm_nt(a, b, L.EPI_ATOMIC_F32, out=out, splitk=4)
        report(f"x3 atomic split-K {name}", out, acc, rtol=0, atol=at)


@pytest.mark.parametrize("M,N,K", [(25216, 512, 2048), (25216, 1536, 512), (128, 1024, 1024), (6272 + 17, 256, 64)])
def test_gemm_tn_x3_vs_fp64(mode, M, N, K):
    from maed_amd import ops
    name, tol = mode
    Y, X = rnd(M, N, seed=12), rnd(M, K, seed=13)
    dW0, db0 = rnd(N, K, seed=14), rnd(N, seed=15)
    dW, db = dW0.to(DEV), db0.to(DEV)
    ops.gemm_tn_wgrad(Y.to(DEV), X.to(DEV), dW=dW, dbias=db)
    ref = Y.double().t() @ X.double()
    report(f"gemm_tn {name} [{M}x{N}x{K}] dW", dW.double().cpu() - dW0.double(), ref, rtol=0, atol=tol * ref.abs().max().item() + 1e-5)
    report(f"gemm_tn {name} [{M}x{N}x{K}] dbias", db, db0.double() + Y.double().sum(0), rtol=1e-5, atol=1e-3)


def _conv_ref(x, w, stride):
    H, W = x.shape[-2:]
    Ho, Wo = -(-H // stride), -(-W // stride)
    ph, pw = max((Ho - 1) * stride + 3 - H, 0), max((Wo - 1) * stride + 3 - W, 0)
    return F.conv2d(F.pad(x.double(), [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]), w.double(), None, stride)


@pytest.mark.parametrize("stride,H,Cin,Cout", [(1, 56, 64, 64), (2, 56, 128, 128), (1, 14, 256, 256), (2, 28, 256, 256)])
def test_conv3x3_x3_fwd_dgrad_wgrad(mode, stride, H, Cin, Cout):
    """the backbone's four 3x3 layer shapes (resnetv2.py:74-93) on 4 frames: forward (+ GroupNorm statistics), stride-1 input gradient from the
    transposed image, stride-1 weight gradient"""
    from maed_amd import ops
    name, tol = mode
    N = 3                                                   # (14 x 14 x 3 = 588 rows: the weight-gradient kernel's ragged last tile)
    x = rnd(N, Cin, H, H, seed=30).contiguous(memory_format=torch.channels_last)
    w = rnd(Cout, Cin, 3, 3, seed=31, scale=(9 * Cin) ** -0.5)
    ref = _conv_ref(x, w, stride)
    sums = torch.zeros(N, 32, 2, dtype=torch.float64, device=DEV)
    y = ops.conv3x3(x.to(DEV), w.permute(0, 2, 3, 1).contiguous().to(DEV), stride, gn_sums=sums if ref.shape[-1] ** 2 >= 128 else None)
    report(f"conv3x3 {name} s{stride} {H}x{H} {Cin}->{Cout}", y, ref, rtol=0, atol=tol * ref.abs().max().item())
    if ref.shape[-1] ** 2 >= 128:
        g = y.double().cpu().permute(0, 2, 3, 1).reshape(N, -1, 32, Cout // 32)
        report(f"conv3x3 {name} GroupNorm sums", sums, torch.stack([g.sum((1, 3)), (g * g).sum((1, 3))], -1), rtol=1e-5, atol=1e-2)
    if stride == 1:
        xg, wg = x.double().requires_grad_(True), w.double().requires_grad_(True)
        dy = rnd(*ref.shape, seed=32).contiguous(memory_format=torch.channels_last)
        gx, gw = torch.autograd.grad(F.conv2d(xg, wg, None, 1, 1), (xg, wg), dy.double())
        dx = ops.conv3x3(dy.to(DEV), w.permute(2, 3, 1, 0).contiguous().to(DEV), 1, w_layout=1)
        report(f"conv3x3 {name} dgrad", dx, gx, rtol=0, atol=tol * gx.abs().max().item())
        dW = ops.conv3x3_wgrad(dy.to(DEV), x.to(DEV))
        report(f"conv3x3 {name} wgrad", dW.permute(0, 3, 1, 2), gw, rtol=0, atol=tol * gw.abs().max().item())


def test_conv1x1_x3_groupnorm_statistics(mode):
    from maed_amd import ops, _lib as L
    name, tol = mode
    Fr, hw, Cin, Cout = 3, 56 * 56, 64, 256
    M = Fr * hw
    x, w = rnd(M, Cin, seed=20), rnd(Cout, Cin, seed=21, scale=Cin ** -0.5)
    sums = torch.zeros(Fr, 32, 2, dtype=torch.float64, device=DEV)
    y = torch.empty(M, Cout, device=DEV)
    xd, wd = x.to(DEV), w.to(DEV)
    L.check(L.lib().maed_conv1x1_fwd(xd.data_ptr(), Cin, wd.data_ptr(), Cin, M, Cout, Cin, y.data_ptr(), Cout, hw, sums.data_ptr(), L.F32,
                                     torch.cuda.current_stream().cuda_stream), "conv1x1")
    ref = x.double() @ w.double().t()
    report(f"conv1x1 {name}", y, ref, rtol=0, atol=tol * ref.abs().max().item())
    g = y.double().cpu().view(Fr, hw, 32, Cout // 32)
    report(f"conv1x1 {name} GroupNorm sums", sums, torch.stack([g.sum((1, 3)), (g * g).sum((1, 3))], -1), rtol=1e-5, atol=1e-2)


@pytest.mark.parametrize("Fr,P,H", [(3, 197, 8), (2, 257, 2), (1, 600, 1), (2, 5, 2)])
def test_attn_spatial_x3_fwd_bwd_vs_fp64(mode, Fr, P, H):
    """fp32 q/k/v, split-bf16 contractions (csrc/attn_x3.hip; vision_transformer.py:206-214) against fp64 autograd through the oracle"""
    from maed_amd import ops
    from oracle import maed_ref as R
    name, tol = mode
    qkv = rnd(Fr, P, 3 * 64 * H, seed=P)
    do = rnd(Fr, P, 64 * H, seed=4)
    x = qkv.double().requires_grad_(True)
    qq, kk, vv = R.split_qkv(x, H)
    oref = R.attention_spatial(qq, kk, vv, 64 ** -0.5)
    lse_ref = torch.logsumexp((qq @ kk.transpose(-2, -1)) * 64 ** -0.5, dim=-1)
    oref.backward(do.double())
    o, lse = ops.attn_spatial_fwd(qkv.to(DEV), H)
    dqkv = ops.attn_spatial_bwd(qkv.to(DEV), o, do.to(DEV), lse, H)
    report(f"attn_spatial {name} fwd [F{Fr} P{P} H{H}]", o, oref.detach(), rtol=0, atol=2 * tol * oref.abs().max().item())
    report(f"attn_spatial {name} lse", lse, lse_ref.detach(), rtol=0, atol=20 * tol)
    report(f"attn_spatial {name} bwd", dqkv, x.grad, rtol=0, atol=4 * tol * x.grad.abs().max().item())


@pytest.mark.parametrize("N,T,P,H", [(8, 16, 197, 8), (2, 16, 9, 2), (3, 8, 5, 1), (1, 32, 33, 2)])
def test_attn_temporal_x3_fwd_bwd_vs_fp64(N, T, P, H):
    """temporal attention (vision_transformer.py:216-228) on fp32 operands with split-bf16 contractions (attn_tm_x3_fwd / _bwd: one-tile virtual sequences,
    bf16x3 only) at the cfg3 shape and at ragged token counts, against fp64 autograd through the oracle; the split kernel must really be the one that ran"""
    from maed_amd import ops
    from oracle import maed_ref as R
    tol = 1e-4
    Fr = N * T
    qkv = rnd(Fr, P, 3 * 64 * H, seed=P)
    do = rnd(Fr, P, 64 * H, seed=4)
    x = qkv.double().requires_grad_(True)
    qq, kk, vv = R.split_qkv(x, H)
    oref = R.attention_temporal(qq, kk, vv, T, 64 ** -0.5)
    oref.backward(do.double())
    o, lse = ops.attn_temporal_fwd(qkv.to(DEV), H, T, prec="bf16x3")
    dqkv = ops.attn_temporal_bwd(qkv.to(DEV), o, do.to(DEV), lse, H, T, prec="bf16x3")
    acc = ops.attn_temporal_bwd(qkv.to(DEV), o, do.to(DEV), lse, H, T, dqkv=dqkv.clone(), accumulate=True, prec="bf16x3")
    o_exact, lse_exact = ops.attn_temporal_fwd(qkv.to(DEV), H, T)
    assert not torch.equal(o, o_exact), "the split kernel did not run"
    report(f"attn_temporal bf16x3 fwd [N{N} T{T} P{P} H{H}]", o, oref.detach(), rtol=0, atol=2 * tol * oref.abs().max().item())
    report("attn_temporal bf16x3 lse vs exact kernel", lse, lse_exact, rtol=0, atol=20 * tol)
    report("attn_temporal bf16x3 bwd", dqkv, x.grad, rtol=0, atol=4 * tol * x.grad.abs().max().item())
    report("attn_temporal bf16x3 bwd accumulate", acc, 2 * x.grad, rtol=0, atol=8 * tol * x.grad.abs().max().item())

