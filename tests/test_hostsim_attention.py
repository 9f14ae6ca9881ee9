"""Attention kernels (maed_amd/csrc/attn_spatial.hip incl. the MFMA forward and both MFMA backward passes, attn_temporal.hip)
and the whole fused STE Block (block.hip: LayerNorm, GEMMs, both attentions, attentive addition, MLP -- 11 launches forward,
~20 backward) on the host simulator against fp64 autograd through the CPU oracle.  Same comparisons as the `-m gpu` suite at
sizes a CPU finishes in seconds; the simulator emulates the MFMA fragment layouts lane for lane, so a wrong layout, swizzle
or k-slot permutation fails HERE."""
import pytest
import torch

from oracle import maed_ref as R
from maed_amd import _lib as L
from maed_amd import ops

from _hostsim import patched
from _util import q, rnd, tol

CASES = [("f32-valu", torch.float32, 1), ("bf16-valu", torch.bfloat16, 1), ("bf16-mfma", torch.bfloat16, 2)]


def close(got, ref, rtol, atol):
    assert torch.allclose(got.double(), ref.double(), rtol=rtol, atol=atol), (got.double() - ref.double()).abs().max().item()


@pytest.mark.parametrize("name,dtype,impl", CASES)
@pytest.mark.parametrize("Fr,P,H", [(2, 5, 2), (1, 70, 1), (1, 197, 1)])
def test_attn_spatial_fwd_bwd(name, dtype, impl, Fr, P, H):
    if P == 197 and impl != 2:
        pytest.skip("the long case is for the MFMA path (7 key tiles, masked last tile)")
    qkv = q(rnd(Fr, P, 3 * 64 * H, seed=P), dtype)
    do = q(rnd(Fr, P, 64 * H, seed=4), dtype)
    x = qkv.double().requires_grad_(True)
    qq, kk, vv = R.split_qkv(x, H)
    oref = R.attention_spatial(qq, kk, vv, 64 ** -0.5)
    lse_ref = torch.logsumexp((qq @ kk.transpose(-2, -1)) * 64 ** -0.5, dim=-1)
    oref.backward(do.double())
    with patched():
        o, lse = ops.attn_spatial_fwd(qkv.to(dtype), H, impl)
        dqkv = ops.attn_spatial_bwd(qkv.to(dtype), o, do.to(dtype), lse, H, impl=impl)
    t = tol(dtype)
    close(o.float(), oref.detach(), **t)
    close(lse, lse_ref.detach(), rtol=1e-4, atol=1e-3 if dtype == torch.float32 else 2e-2)
    close(dqkv.float(), x.grad, **tol(dtype, 0.5))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,T,P,H", [(2, 3, 5, 2), (1, 16, 9, 1), (1, 64, 3, 1)])       # bf16: one-tile backward (T <= 32) and the general one (T = 64)
def test_attn_temporal_fwd_bwd(dtype, N, T, P, H):
    Fr = N * T
    qkv = q(rnd(Fr, P, 3 * 64 * H, seed=5), dtype)
    do = q(rnd(Fr, P, 64 * H, seed=6), dtype)
    x = qkv.double().requires_grad_(True)
    qq, kk, vv = R.split_qkv(x, H)
    oref = R.attention_temporal(qq, kk, vv, T, 64 ** -0.5)
    oref.backward(do.double())
    with patched():
        o, lse = ops.attn_temporal_fwd(qkv.to(dtype), H, T)
        dqkv = ops.attn_temporal_bwd(qkv.to(dtype), o, do.to(dtype), lse, H, T)
    close(o.float(), oref.detach(), **tol(dtype))
    close(dqkv.float(), x.grad, **tol(dtype, 0.5))


@pytest.mark.parametrize("dtype,f32_mode,rows", [(torch.float32, "exact", 9), (torch.float32, "bf16x3", 20), (torch.float32, "bf16x6", 9), (torch.bfloat16, "exact", 9),
                                                 (torch.bfloat16, "exact", 20), (torch.float32, "bf16x3+bwd:bf16x1", 20), (torch.float32, "bf16x3+bwd:bf16", 20),
                                                 (torch.float32, "bf16x3+bwd:bf16/fp32 operands", 20)])
def test_ste_block_forward_backward_vs_oracle(dtype, f32_mode, rows):
    """one whole Block through maed_ste_block_fwd/bwd (vision_transformer.py:244-261) incl. every parameter gradient: f32 parity mode (exact VALU
    kernels + transposed copies), f32 on the split-bf16 MFMA kernels (bf16x3 / bf16x6: the bf16 mode's kernel sequence on fp32 operands) and the
    bf16 MFMA throughput mode; 20 tokens x 2 frames = 40 rows = 2 LayerNorm workgroups for the dgamma / dbeta partials"""
    impl = 0
    from functools import partial
    import torch.nn as nn
    from maed_amd.vision_transformer import Block
    N, T, P, H = 1, 2, rows, 2
    C, Fr = 64 * H, N * T
    p = {k[len("encoder.blocks.0."):]: v for k, v in R.make_params(embed_dim=C, depth=1, hidden_dim=64, layers=(1, 1, 1), n_tokens=P, seed=3).items()
         if k.startswith("encoder.blocks.0.")}
    p = {k: v * (3.0 if k.endswith("weight") and v.dim() == 2 else 1.0) for k, v in p.items()}
    x, dy = rnd(Fr, P, C, seed=1), rnd(Fr, P, C, seed=2)
    pd = {k: v.double().requires_grad_(True) for k, v in p.items()}
    xr = x.double().requires_grad_(True)
    yref = R.block(xr, pd, "", H, T)
    yref.backward(dy.double())
    blk = Block(C, H, mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), st_mode="parallel", compute_dtype=dtype, impl=impl)
    blk.load_state_dict(p)
    xg = x.clone().requires_grad_(True)
    old = ops.get_float32_matmul_precision()
    f32_mode, _, bwd = f32_mode.partition("+bwd:")          # "bf16x3+bwd:bf16x1" = the mixed mode of round 4: split forward, one-plane backward products;
                                                            # "+bwd:bf16" (round 5): split forward into an fp32 work buffer, bf16 twins saved, the bf16 mode's backward on them
    bwd, _, no_planes = bwd.partition("/")                  # default: fc1's activation stored as (hi, lo) planes, fc2 on the plane
                                                            # kernel (MAED_OPT_X3_PLANES); "/fp32 operands": the fp32-operand kernels + a cast pass for every twin
    try:
        ops.set_float32_matmul_precision(f32_mode)
        ops.set_float32_backward_precision(bwd or None)
        twins = ops.TWIN_FORWARDS[0]
        with patched() as lib:
            old_planes = lib.maed_get_option(L.OPT_X3_PLANES)
            if no_planes:
                lib.maed_set_option(L.OPT_X3_PLANES, 0)
            try:
                y = blk(xg, T)
                y.backward(dy)
            finally:
                lib.maed_set_option(L.OPT_X3_PLANES, old_planes)
        assert (ops.TWIN_FORWARDS[0] - twins == 1) == (bwd == "bf16")
    finally:
        ops.set_float32_matmul_precision(old)
        ops.set_float32_backward_precision(None)
    f32 = dtype == torch.float32
    tl = (dict(rtol=3e-4, atol=3e-4) if f32_mode == "bf16x3" else dict(rtol=1e-4, atol=1e-4)) if f32 else dict(rtol=3e-2, atol=3e-2)
    close(y.detach(), yref.detach(), **tl)                  # (mixed mode: the FORWARD keeps the split engine's tolerance)
    if bwd:
        f32, tl = False, dict(rtol=3e-2, atol=3e-2)         # ... its gradients are held to the bf16 mode's
    close(xg.grad, xr.grad, **tl)
    for name, prm in blk.named_parameters():
        ref = pd[name].grad
        scale = max(ref.abs().max().item(), 1e-3)
        assert prm.grad is not None, name
        close(prm.grad, ref, rtol=1e-3 if f32 else 5e-2, atol=(1e-4 if f32 else 3e-2) * scale)


def test_twin_forward_with_plane_storage_of_fc1s_activation_changes_no_bit():
    """MAED_OPT_X3_PLANES (round 5): fc1's activation stored as (hi, lo) bf16 planes + fc2 on the plane kernel (csrc/gemm_x3p.hip, every tile variant) against the fp32
    activation + twin + fp32-operand kernel: the same output, and the input gradient and every parameter gradient bit for bit"""
    from functools import partial
    import torch.nn as nn
    from maed_amd.vision_transformer import Block
    from _hostsim import option
    C, H, T, P = 128, 2, 2, 12
    torch.manual_seed(5)
    blk = Block(C, H, mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), st_mode="parallel", compute_dtype=torch.float32, impl=0)
    x, dy = rnd(T, P, C, seed=1), rnd(T, P, C, seed=2)
    old = ops.get_float32_matmul_precision()
    res = {}
    try:
        ops.set_float32_matmul_precision("bf16x3")
        ops.set_float32_backward_precision("bf16")
        with patched() as lib:
            for variant, ln in ((0, 0), (6, 0), (5, 1)):       # (fc2's kernel variant, LayerNorm outputs as planes + qkv / fc1 on the plane kernel); the other tile
                                                                # variants are held to bit equality per product in tests/test_hostsim_gemm.py
                with option(lib, L.OPT_X3_PLANES, variant), option(lib, L.OPT_X3_PLANES_LN, ln):
                    blk.zero_grad()
                    xg = x.clone().requires_grad_(True)
                    y = blk(xg, T)
                    y.backward(dy)
                    res[variant, ln] = [y.detach().clone(), xg.grad.clone()] + [prm.grad.clone() for prm in blk.parameters()]
    finally:
        ops.set_float32_matmul_precision(old)
        ops.set_float32_backward_precision(None)
    base = res[0, 0]
    for key, got in res.items():
        # the output: same products, but at this size (one or two output tiles, K up to 512) the fp32-operand route may split K over workgroups -- another summation
        # order (tests/test_hostsim_gemm.py holds the two kernels to bit equality where both run unsplit)
        assert (base[0] - got[0]).abs().max() <= 2e-6 * base[0].abs().max(), key
        if key[1] == 0:
            for a, b in zip(base[1:], got[1:]):      # fc2 only: the backward reads the hi plane = the twin and fc1's bf16 pre-activation: the same bits
                assert torch.equal(a, b), key
        else:
            # qkv / fc1 on the plane kernel: their results feed what the backward reads (qkv, the attention outputs, fc1's pre-activation).  At 40 rows the
            # fp32-operand route is not the split kernel the plane kernel mirrors, so a sum differs in its last bit here and there and the bf16 twin of it by one
            # ulp: gradients agree to a few 1e-3 (measured 2.8e-3 at most), far inside what test_ste_block_forward_backward_vs_oracle allows either path
            for a, b in zip(base[1:], got[1:]):
                assert (a - b).abs().max() <= 6e-3 * a.abs().max() + 1e-6, key


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_block_forward_without_backward_takes_the_inference_entry_point_and_equals_the_training_forward(dtype):
    """under torch.no_grad the Block runs maed_ste_block_infer (fc1's pre-activation, which only GELU' reads, is not stored): same output bit for bit"""
    from functools import partial
    import torch.nn as nn
    from maed_amd.vision_transformer import Block
    torch.manual_seed(0)
    C, H, T, P = 128, 2, 2, 7
    blk = Block(C, H, mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), st_mode="parallel", compute_dtype=dtype, impl=0)
    x = rnd(T, P, C, seed=1)
    with patched() as h:
        calls = {"fwd": 0, "infer": 0}
        real_f, real_i = h.maed_ste_block_fwd, h.maed_ste_block_infer

        class Spy:
            def __init__(self, fn, key):
                self.fn, self.key = fn, key

            def __call__(self, *a):
                calls[self.key] += 1
                return self.fn(*a)
        h.maed_ste_block_fwd, h.maed_ste_block_infer = Spy(real_f, "fwd"), Spy(real_i, "infer")
        try:
            y_train = blk(x.clone().requires_grad_(True), T)
            with torch.no_grad():
                y_eval = blk(x, T)
        finally:
            h.maed_ste_block_fwd, h.maed_ste_block_infer = real_f, real_i
    assert calls == {"fwd": 1, "infer": 1}, calls
    assert torch.equal(y_train.detach(), y_eval)


def test_residual_gradient_hand_off_between_blocks_is_used_and_a_bypass_is_loud():
    """bf16 blocks hand the compute-dtype copy of their residual gradient to the previous block (ops._TWIN, keyed on tensor identity).  Two training steps of a
    two-block chain: the hand-off is used on every second block backward and nothing warns (the copy the LAST block leaves is stale, not a bypass); a hook
    that replaces the gradient between the blocks is a real bypass: same result through the extra cast pass, one RuntimeWarning."""
    import warnings
    from functools import partial
    import torch.nn as nn
    from maed_amd.vision_transformer import Block
    torch.manual_seed(0)
    C, H, T, P = 128, 2, 2, 5
    blocks = [Block(C, H, mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), st_mode="parallel", compute_dtype=torch.bfloat16, impl=0) for _ in range(2)]
    for i, b in enumerate(blocks):
        b._chain_index = i
    x = rnd(T, P, C, seed=1)
    dy = rnd(T, P, C, seed=2)

    def run(hook):
        for b in blocks:
            for prm in b.parameters():
                prm.grad = None
        xg = x.clone().requires_grad_(True)
        h = blocks[0](xg, T)
        if hook:
            h.register_hook(lambda g: g.clone())
        blocks[1](h, T).backward(dy)
        return xg.grad.clone()

    ops._TWIN_WARNED[0] = False
    with patched():
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            h0 = list(ops.TWIN_HITS)
            g1 = run(False)
            g2 = run(False)
            assert not [m for m in w if "hand-off" in str(m.message)], [str(m.message) for m in w]
            assert ops.TWIN_HITS[0] - h0[0] == 2 and ops.TWIN_HITS[1] - h0[1] == 2, (h0, ops.TWIN_HITS)
            assert torch.equal(g1, g2)
            g3 = run(True)
            assert len([m for m in w if "hand-off" in str(m.message) and issubclass(m.category, RuntimeWarning)]) == 1
            run(True)
            assert len([m for m in w if "hand-off" in str(m.message)]) == 1        # once per process
    assert torch.allclose(g3, g1, rtol=2e-2, atol=2e-2 * float(g1.abs().max()))
    ops._TWIN_WARNED[0] = False


# ---- long-sequence (K/V-tiled) kernels: attn_long.hip ------------------------------------------------------------------
@pytest.mark.parametrize("Fr,L_,H,impl", [
    (2, 5, 2, L.IMPL_MFMA_LONG),          # one partial tile, one partially filled wave
    (1, 197, 2, L.IMPL_MFMA_LONG),        # spatial shape of cfg3 through the tiled kernels: 2 row tiles x 4 streamed tiles, ragged ends
    (1, 64, 1, L.IMPL_MFMA_LONG),         # exactly one full tile (no masking anywhere)
])    # (AUTO picking the tiled kernels beyond the whole-head limits: the 560-token coupling test below and scripts/check_new_paths.py)
def test_attn_long_fwd_bwd(Fr, L_, H, impl):
    dtype = torch.bfloat16
    qkv = q(rnd(Fr, L_, 3 * 64 * H, seed=L_), dtype)
    do = q(rnd(Fr, L_, 64 * H, seed=4), dtype)
    x = qkv.double().requires_grad_(True)
    qq, kk, vv = R.split_qkv(x, H)
    oref = R.attention_spatial(qq, kk, vv, 64 ** -0.5)
    lse_ref = torch.logsumexp((qq @ kk.transpose(-2, -1)) * 64 ** -0.5, dim=-1)
    oref.backward(do.double())
    with patched():
        o, lse = ops.attn_spatial_fwd(qkv.to(dtype), H, impl)
        dqkv = ops.attn_spatial_bwd(qkv.to(dtype), o, do.to(dtype), lse, H, impl=impl)
        acc = ops.attn_spatial_bwd(qkv.to(dtype), o, do.to(dtype), lse, H, dqkv=dqkv.clone(), accumulate=True, impl=impl)
    close(o.float(), oref.detach(), **tol(dtype))
    close(lse, lse_ref.detach(), rtol=1e-4, atol=2e-2)
    close(dqkv.float(), x.grad, **tol(dtype, 0.5))
    close(acc.float(), 2 * x.grad, **tol(dtype, 1.0))            # accumulate=1 adds into the existing gradient


def test_attn_long_agrees_with_whole_head_kernels():
    """same inputs through the whole-head MFMA kernels and the tiled ones: identical math up to the online-softmax rescaling order"""
    dtype = torch.bfloat16
    qkv = q(rnd(2, 150, 3 * 64, seed=9), dtype).to(dtype)
    do = q(rnd(2, 150, 64, seed=10), dtype).to(dtype)
    with patched():
        o1, l1 = ops.attn_spatial_fwd(qkv, 1, L.IMPL_MFMA)
        o2, l2 = ops.attn_spatial_fwd(qkv, 1, L.IMPL_MFMA_LONG)
        g1 = ops.attn_spatial_bwd(qkv, o1, do, l1, 1, impl=L.IMPL_MFMA)
        g2 = ops.attn_spatial_bwd(qkv, o1, do, l1, 1, impl=L.IMPL_MFMA_LONG)
    close(o2.float(), o1.float(), rtol=1e-2, atol=1e-2)
    close(l2, l1, rtol=1e-5, atol=1e-5)
    close(g2.float(), g1.float(), rtol=1e-2, atol=1e-2)


def test_coupling_mode_at_a_sequence_beyond_the_whole_head_limit():
    """st_mode='coupling' with T*P = 8 * 70 = 560 tokens per clip (vision_transformer.py:160-163): the reshape-free view of the
    qkv rows + the tiled kernels against the oracle's explicit reshape_T formulation, forward and backward"""
    from maed_amd import ste_modes
    N, T, P, H = 1, 8, 70, 1
    dtype = torch.bfloat16
    qkv = q(rnd(N * T, P, 3 * 64 * H, seed=21), dtype)
    do = q(rnd(N * T, P, 64 * H, seed=22), dtype)
    x = qkv.double().requires_grad_(True)
    oref = R.attention_coupling(*R.split_qkv(x, H), T, 64 ** -0.5)
    oref.backward(do.double())
    xg = qkv.to(dtype).requires_grad_(True)
    with patched():
        o = ste_modes.SpatialAttnFn.apply(xg.view(N, T * P, 3 * 64 * H), H, L.IMPL_AUTO).view(N * T, P, 64 * H)
        o.backward(do.to(dtype))
    close(o.float(), oref.detach(), **tol(dtype))
    close(xg.grad.float(), x.grad, **tol(dtype, 0.5))


@pytest.mark.parametrize("dtype,impl", [(torch.float32, L.IMPL_AUTO), (torch.bfloat16, L.IMPL_VALU)])
def test_attn_long_exact_valu_kernels(dtype, impl):
    """f32 parity mode (and bf16 with the VALU kernels forced) past the whole-head LDS limit of ~310 tokens: tiled exact kernels"""
    Fr, L_, H = 1, 330, 2
    qkv = q(rnd(Fr, L_, 3 * 64 * H, seed=L_), dtype)
    do = q(rnd(Fr, L_, 64 * H, seed=4), dtype)
    x = qkv.double().requires_grad_(True)
    qq, kk, vv = R.split_qkv(x, H)
    oref = R.attention_spatial(qq, kk, vv, 64 ** -0.5)
    lse_ref = torch.logsumexp((qq @ kk.transpose(-2, -1)) * 64 ** -0.5, dim=-1)
    oref.backward(do.double())
    with patched():
        o, lse = ops.attn_spatial_fwd(qkv.to(dtype), H, impl)
        dqkv = ops.attn_spatial_bwd(qkv.to(dtype), o, do.to(dtype), lse, H, impl=impl)
        acc = ops.attn_spatial_bwd(qkv.to(dtype), o, do.to(dtype), lse, H, dqkv=dqkv.clone(), accumulate=True, impl=impl)
    close(o.float(), oref.detach(), **tol(dtype))
    close(lse, lse_ref.detach(), rtol=1e-4, atol=1e-3 if dtype == torch.float32 else 2e-2)
    close(dqkv.float(), x.grad, **tol(dtype, 0.5))
    close(acc.float(), 2 * x.grad, **tol(dtype, 1.0))


def test_twin_forward_at_256_channels_and_few_rows_leaves_no_twin_unwritten(monkeypatch):
    """with C >= 256 and a few dozen rows the fp32 STORE products used to take a split-K route whose atomic epilogue writes the fp32 result only -- the qkv
    twin the bf16 backward reads stayed uninitialised.  Every arena / work buffer is handed out NaN-filled here: a field nobody wrote shows up as a NaN gradient."""
    from functools import partial
    import torch.nn as nn
    from maed_amd.vision_transformer import Block
    C, H, T, P = 256, 4, 2, 12
    torch.manual_seed(7)
    blk = Block(C, H, mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), st_mode="parallel", compute_dtype=torch.float32, impl=0)
    x, dy = rnd(T, P, C, seed=1), rnd(T, P, C, seed=2)
    real = ops._aligned_bytes

    def poisoned(nbytes, device, align=256):
        buf = real(nbytes, device, align) if align != 256 else real(nbytes, device)
        buf.fill_(0xFF)                   # 0xFFFF = a bf16 NaN, 0xFFFFFFFF = an fp32 NaN
        return buf

    monkeypatch.setattr(ops, "_aligned_bytes", poisoned)
    monkeypatch.setattr(ops, "_SCRATCH", {})
    old = ops.get_float32_matmul_precision()
    try:
        ops.set_float32_matmul_precision("bf16x3")
        ops.set_float32_backward_precision("bf16")
        twins = ops.TWIN_FORWARDS[0]
        with patched():
            xg = x.clone().requires_grad_(True)
            y = blk(xg, T)
            y.backward(dy)
        assert ops.TWIN_FORWARDS[0] - twins == 1
    finally:
        ops.set_float32_matmul_precision(old)
        ops.set_float32_backward_precision(None)
    assert torch.isfinite(y).all() and torch.isfinite(xg.grad).all()
    for name, prm in blk.named_parameters():
        assert prm.grad is not None and torch.isfinite(prm.grad).all(), name
    # and the gradients are the oracle's, to the bf16 backward's tolerance
    pd = {k: v.detach().double().requires_grad_(True) for k, v in blk.state_dict().items()}
    xr = x.double().requires_grad_(True)
    R.block(xr, pd, "", H, T).backward(dy.double())
    close(xg.grad, xr.grad, rtol=3e-2, atol=3e-2)
    for name, prm in blk.named_parameters():
        ref = pd[name].grad
        close(prm.grad, ref, rtol=5e-2, atol=3e-2 * max(ref.abs().max().item(), 1e-3))
