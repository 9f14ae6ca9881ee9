"""The bf16 MFMA GEMM kernels (maed_amd/csrc/gemm.hip, gemm_tn.hip) on the host simulator: the x86 build of the same sources
with v_mfma_f32_32x32x16_bf16, global_load_lds_dwordx4, v_perm_b32 and the LDS swizzles emulated lane for lane
(tests/hostsim/hip/hip_runtime.h).  Checks fragment layouts, swizzle algebra, the LDS-shuffled epilogue, ragged edges and
every fused epilogue against torch on the bf16-rounded operands -- without a GPU."""
import pytest
import torch

from maed_amd import _lib as L
from maed_amd import ops

from _hostsim import option, patched


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def gelu(x):
    return 0.5 * x * (1 + torch.erf(x / 2 ** 0.5))


def dgelu(x):
    return 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5


@pytest.mark.parametrize("impl", [L.IMPL_MFMA, 3, 4])      # register-staged, direct-to-LDS with one / two buffers
@pytest.mark.parametrize("M,N,K", [(130, 136, 128), (64, 256, 64)])
def test_gemm_nt_store_variants(impl, M, N, K):
    A, B, bias = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2, scale=K ** -0.5).bfloat16(), rnd(N, seed=3)
    ref = A.float() @ B.float().t() + bias
    with patched():
        out = ops.gemm_nt(A, B, L.EPI_STORE, bias=bias, impl=impl)
    assert torch.allclose(out.float(), ref, rtol=2e-2, atol=2e-2)
    assert (out.float() - ref.bfloat16().float()).abs().max() <= 2 * 2 ** -8 * ref.abs().max()


@pytest.mark.parametrize("epi", ["gelu", "resid", "dgelu", "store_f32", "tanh", "add"])
def test_gemm_nt_fused_epilogues(epi):
    M, N, K = 96, 200, 128                      # ragged M tile, N not a multiple of 8*... (exercises the scalar epilogue tail)
    A, B, bias = rnd(M, K, seed=4).bfloat16(), rnd(N, K, seed=5, scale=K ** -0.5).bfloat16(), rnd(N, seed=6)
    acc = A.float() @ B.float().t()
    with patched():
        if epi == "gelu":
            out, pre = ops.gemm_nt(A, B, L.EPI_GELU, bias=bias)
            want_pre = (acc + bias).bfloat16()
            assert torch.allclose(pre.float(), want_pre.float(), atol=2e-2)
            assert torch.allclose(out.float(), gelu(pre.float()), rtol=2e-2, atol=2e-2)     # activation of the STORED pre-activation
        elif epi == "resid":
            aux = rnd(M, N, seed=7)
            out = ops.gemm_nt(A, B, L.EPI_RESID_F32, bias=bias, aux=aux)
            assert out.dtype == torch.float32 and torch.allclose(out, aux + acc + bias, rtol=1e-4, atol=1e-4)
        elif epi == "dgelu":
            aux = rnd(M, N, seed=8).bfloat16()
            out = ops.gemm_nt(A, B, L.EPI_MUL_DGELU, aux=aux)
            assert torch.allclose(out.float(), acc * dgelu(aux.float()), rtol=3e-2, atol=3e-2)
        elif epi == "store_f32":
            out = ops.gemm_nt(A, B, L.EPI_STORE_F32, bias=bias)
            assert torch.allclose(out, acc + bias, rtol=1e-4, atol=1e-4)
        elif epi == "tanh":
            out = ops.gemm_nt(A, B, L.EPI_TANH, bias=bias)
            assert torch.allclose(out.float(), torch.tanh(acc + bias), atol=1e-2)
        else:
            aux = rnd(M, N, seed=9).bfloat16()
            out = ops.gemm_nt(A, B, L.EPI_ADD, aux=aux)
            assert torch.allclose(out.float(), acc + aux.float(), rtol=2e-2, atol=2e-2)


def test_gemm_nt_splitk_atomic():
    M, N, K = 70, 128, 256
    A, B = rnd(M, K, seed=10).bfloat16(), rnd(N, K, seed=11, scale=K ** -0.5).bfloat16()
    with patched():
        out = ops.gemm_nt(A, B, L.EPI_ATOMIC_F32, splitk=2)
    assert torch.allclose(out, A.float() @ B.float().t(), rtol=1e-4, atol=1e-4)


# (M % 32 == 0: the LDS-DMA + transposing-read kernel of round 5, csrc/gemm_tn2.hip -- one tile, exactly the ring depth, more tiles than stages, several M-splits,
#  column tiles that end inside a 128-block, a second K tile (rotating bias owner); the others: the register-transposing kernel incl. its ragged tile)
@pytest.mark.parametrize("M,N,K", [(200, 136, 72), (64, 128, 128), (37, 8, 8), (130, 64, 256), (32, 128, 128), (128, 136, 72), (160, 8, 8), (416, 264, 200),
                                   (2048, 64, 136)])
@pytest.mark.parametrize("dma", [2, 3, 0])         # 2 / 3: the round-5 kernel's 128 x 128 / 256 x 256 tile forced for every shape, 0: the register-transposing kernel
def test_gemm_tn_wgrad_and_bias(M, N, K, dma):
    from _hostsim import option
    Y, X = rnd(M, N, seed=12).bfloat16(), rnd(M, K, seed=13).bfloat16()
    dW0, db0 = rnd(N, K, seed=14), rnd(N, seed=15)
    dW, db = dW0.clone(), db0.clone()
    with patched() as lib, option(lib, L.OPT_TN_DMA, dma):
        ops.gemm_tn_wgrad(Y, X, dW=dW, dbias=db)          # ACCUMULATES into dW / dbias
    # transpose-detecting: Y and X are unrelated random matrices, N != K in most cases
    assert torch.allclose(dW, dW0 + Y.float().t() @ X.float(), rtol=1e-4, atol=1e-3)
    assert torch.allclose(db, db0 + Y.float().sum(0), rtol=1e-4, atol=1e-3)


def test_gemm_tn_dma_strided_operands_and_no_bias():
    """operands that are column slices of wider matrices (row strides 3C / hidden as in the STE block), no bias gradient"""
    from _hostsim import option
    M, N, K = 192, 64, 96
    Yw, Xw = rnd(M, 3 * N, seed=16).bfloat16(), rnd(M, K + 40, seed=17).bfloat16()
    Y, X = Yw[:, N:2 * N], Xw[:, 8:8 + K]
    dW = torch.zeros(N, K)
    for which in (2, 3):
        dW.zero_()
        with patched() as lib, option(lib, L.OPT_TN_DMA, which):
            ops.gemm_tn_wgrad(Y, X, dW=dW)
        assert torch.allclose(dW, Y.float().t() @ X.float(), rtol=1e-4, atol=1e-3)


def test_gemm_tn_dma_default_dispatch_takes_the_big_tile_for_outputs_of_512_and_more():
    from _hostsim import option
    M, N, K = 96, 520, 512
    Y, X = rnd(M, N, seed=18).bfloat16(), rnd(M, K, seed=19).bfloat16()
    dW, db = torch.zeros(N, K), torch.zeros(N)
    with patched() as lib, option(lib, L.OPT_TN_DMA, 1):
        ops.gemm_tn_wgrad(Y, X, dW=dW, dbias=db)
    assert torch.allclose(dW, Y.float().t() @ X.float(), rtol=1e-4, atol=1e-3) and torch.allclose(db, Y.float().sum(0), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("M,N,K", [(300, 264, 128), (256, 256, 192), (70, 512, 320)])
def test_gemm_nt_256_pipelined_tiles(M, N, K):
    """csrc/gemm256.hip (256x256x64 tiles, half-tile LDS-DMA ring, staggered wave groups): slot / fragment / swizzle indexing, ragged M
    and N edges, 2, 3 and 5 K tiles (even and odd tile counts: both drains) against torch on the bf16-rounded operands.  The
    simulator lands an LDS-DMA when it is issued, i.e. EARLIER than hardware: a DMA that re-targets a slot too soon after its last
    read shows up here as wrong numbers; landing too late is what the GPU tests are for."""
    A, B, bias = rnd(M, K, seed=21).bfloat16(), rnd(N, K, seed=22, scale=K ** -0.5).bfloat16(), rnd(N, seed=23)
    ref = A.float() @ B.float().t() + bias
    with patched():
        out = ops.gemm_nt(A, B, L.EPI_STORE, bias=bias, impl=L.IMPL_MFMA_256)
    assert (out.float() - ref).abs().max() <= 2 * 2 ** -8 * ref.abs().max() + 1e-2


def test_gemm_nt_256_epilogues_and_transpose_detection():
    M, N, K = 260, 256, 128
    A, B, bias = rnd(M, K, seed=24).bfloat16(), rnd(N, K, seed=25, scale=K ** -0.5).bfloat16(), rnd(N, seed=26)
    acc = A.float() @ B.float().t()
    aux = rnd(M, N, seed=27)
    with patched():
        out = ops.gemm_nt(A, B, L.EPI_RESID_F32, bias=bias, aux=aux, impl=L.IMPL_MFMA_256)
        act, pre = ops.gemm_nt(A, B, L.EPI_GELU, bias=bias, impl=L.IMPL_MFMA_256)
    assert torch.allclose(out, aux + acc + bias, rtol=1e-4, atol=1e-4)
    assert torch.allclose(pre.float(), (acc + bias).bfloat16().float(), atol=2e-2) and torch.allclose(act.float(), gelu(pre.float()), rtol=2e-2, atol=2e-2)



def test_gemm_nt_f32_few_tiles_long_k_takes_the_split_k_route():
    """fp32, few output tiles, K >= 256 (the decoder tail's GEMMs): bias fill + split-K atomics must equal the plain kernel's answer"""
    M, N, K = 70, 128, 512
    A, B, bias = rnd(M, K, seed=41), rnd(N, K, seed=42, scale=K ** -0.5), rnd(N, seed=43)
    ref = A.double() @ B.double().t() + bias.double()
    with patched():
        out = ops.gemm_nt(A, B, L.EPI_STORE, bias=bias)
        out_nb = ops.gemm_nt(A, B, L.EPI_STORE_F32)
    assert torch.allclose(out.double(), ref, rtol=1e-5, atol=1e-5) and torch.allclose(out_nb.double(), ref - bias.double(), rtol=1e-5, atol=1e-5)



@pytest.mark.parametrize("impl", [L.IMPL_AUTO, L.IMPL_MFMA, L.IMPL_MFMA_256])
def test_gemm_nt_add_epilogue_masks_aux_by_bits(impl):
    """MAED_EPI_ADD with out2 = 1 bit per element of aux (bit c & 7 of byte (r * ldaux + c) / 8: the GroupNorm forward's ReLU mask): the masked residual
    gradient is applied while it is added -- all three epilogue widths (8 / 4 columns, scalar tail)"""
    M, N, K = (260, 256, 128) if impl == L.IMPL_MFMA_256 else (130, 136, 64)
    A, B = rnd(M, K, seed=31).bfloat16(), rnd(N, K, seed=32, scale=K ** -0.5).bfloat16()
    aux = rnd(M, N, seed=33).bfloat16()
    keep = torch.rand(M, N, generator=torch.Generator().manual_seed(34)) > 0.4
    bits = (keep.view(M, N // 8, 8).to(torch.uint8) << torch.arange(8, dtype=torch.uint8)).sum(-1).to(torch.uint8).contiguous()
    ref = A.float() @ B.float().t() + aux.float() * keep
    with patched():
        out = ops.gemm_nt(A, B, L.EPI_ADD, aux=aux, out2=bits, impl=impl)
        out_plain = ops.gemm_nt(A, B, L.EPI_ADD, aux=aux, impl=impl)
    assert torch.allclose(out.float(), ref, rtol=2e-2, atol=2e-2)
    assert torch.allclose(out_plain.float(), A.float() @ B.float().t() + aux.float(), rtol=2e-2, atol=2e-2)


# ---- round 5: the split product on operands stored as (hi, lo) bf16 planes (csrc/gemm_x3p.hip) -------------------------------------------------------------------
def _planes_ref(x):
    hi = x.bfloat16()
    return hi, (x - hi.float()).bfloat16()


@pytest.mark.parametrize("variant", [2, 4, 5, 6, 7])  # 128 x 128 tiles with 2 / 4 stages, 256 x 128 (3 stages), 256 x 256 (2 stages), 128 x 128 with K tiles of 64
@pytest.mark.parametrize("M,N,K", [(130, 136, 96), (128, 128, 32), (64, 264, 160), (257, 72, 64), (300, 392, 64)])      # ragged tiles, one K tile, fewer tiles than stages, more
def test_gemm_nt_planes_is_the_split_product_bit_for_bit(M, N, K, variant):
    A, B, bias = rnd(M, K, seed=31), rnd(N, K, seed=32, scale=K ** -0.5), rnd(N, seed=33)
    with patched():
        Ap, Bp = ops.split_planes(A), ops.split_planes(B)
        for got, want in zip(Ap + Bp, _planes_ref(A) + _planes_ref(B)):
            assert torch.equal(got.view(torch.int16), want.view(torch.int16))
        out, planes, _ = ops.gemm_nt_planes(Ap, Bp, L.EPI_STORE, bias=bias, want_planes=True, variant=variant)
        x3 = ops.gemm_nt(A, B, L.EPI_STORE, bias=bias, prec="bf16x3")
    assert torch.equal(out, x3)                                                   # same K order, same product order as the register-staged kernel
    ref = A.double() @ B.double().t() + bias.double()
    assert (out.double() - ref).abs().max() <= 2 ** -14 * (A.abs().double() @ B.abs().double().t()).max()
    oh, ol = _planes_ref(out)
    assert torch.equal(planes[0].view(torch.int16), oh.view(torch.int16)) and torch.equal(planes[1].view(torch.int16), ol.view(torch.int16))


def test_gemm_nt_planes_epilogues_and_strided_operands():
    M, N, K = 96, 200, 128
    Abig, B, bias, res = rnd(M, 2 * K, seed=41), rnd(N, K, seed=42, scale=K ** -0.5), rnd(N, seed=43), rnd(M, N, seed=44)
    A = Abig[:, K:]                                                                # a column window: leading dimension 2 K
    with patched():
        Ahb, Alb = ops.split_planes(Abig)
        Ap, Bp = (Ahb[:, K:], Alb[:, K:]), ops.split_planes(B)
        act, planes, pre = ops.gemm_nt_planes(Ap, Bp, L.EPI_GELU, bias=bias, want_f32=False, want_planes=True, want_pre=True)
        assert act is None
        a3, p3 = ops.gemm_nt(A.contiguous(), B, L.EPI_GELU, bias=bias, prec="bf16x3")
        resid, _, _ = ops.gemm_nt_planes(Ap, Bp, L.EPI_RESID_F32, bias=bias, aux=res)
        r3 = ops.gemm_nt(A.contiguous(), B, L.EPI_RESID_F32, bias=bias, aux=res, prec="bf16x3")
    hi, lo = _planes_ref(a3)
    assert torch.equal(planes[0].view(torch.int16), hi.view(torch.int16)) and torch.equal(planes[1].view(torch.int16), lo.view(torch.int16))
    assert torch.equal(pre.view(torch.int16), p3.bfloat16().view(torch.int16))
    assert torch.equal(resid, r3)


def test_gemm_nt_planes_rejects_what_it_cannot_do():
    A, B = rnd(32, 48, seed=51), rnd(32, 48, seed=52)
    with patched() as lib:
        Ap, Bp = ops.split_planes(A), ops.split_planes(B)
        with pytest.raises(Exception, match="K %"):
            ops.gemm_nt_planes(Ap, Bp)                                            # K = 48 is not a multiple of 32
        A2, B2 = ops.split_planes(rnd(32, 64, seed=53)), ops.split_planes(rnd(32, 64, seed=54))
        with pytest.raises(Exception, match="epilogue"):
            ops.gemm_nt_planes(A2, B2, L.EPI_TANH)
        with pytest.raises(Exception, match="no output"):
            ops.gemm_nt_planes(A2, B2, want_f32=False)


# ---- round 6: the persistent K-stream kernel (csrc/gemm_sk.hip) --------------------------------------------------------------------------------------
# MAED_OPT_SK_GRID shrinks the grid so that a few workgroups of the simulator see every kind of item: several whole tiles per workgroup (the copy stream
# running through an epilogue), a part whose slab is handed on, a tile finished from one / two slabs, empty ranges.
@pytest.mark.parametrize("M,N,K,grid,mode", [
    (300, 520, 256, 2, 2),      # 2 x 3 tiles (ragged M and N), two workgroups, no K cuts: three whole tiles each, the stream crosses two epilogues
    (300, 520, 256, 4, 3),      # 6 tiles on 4 workgroups: one round whole + two tiles cut... (rounds - 1) * G = 0 whole: all six tiles cut into 4 ranges of 3 pairs
    (256, 512, 384, 3, 3),      # 2 tiles x 3 pairs on 3 workgroups: ranges of 2 pairs: head / (tail + head) / tail
    (256, 256, 512, 3, 3),      # ONE tile cut three ways: the finisher adds two slabs (a middle part)
    (256, 256, 128, 4, 3),      # one pair on four workgroups: three empty ranges
])
def test_gemm_nt_persistent_kstream(M, N, K, grid, mode):
    A, B, bias = rnd(M, K, seed=31).bfloat16(), rnd(N, K, seed=32, scale=K ** -0.5).bfloat16(), rnd(N, seed=33)
    ref = A.float() @ B.float().t() + bias
    with patched() as lib, option(lib, L.OPT_SK, mode), option(lib, L.OPT_SK_GRID, grid):
        out = ops.gemm_nt(A, B, L.EPI_STORE_F32, bias=bias, impl=L.IMPL_MFMA_SK)
        assert torch.allclose(out, ref, rtol=1e-4, atol=1e-4), (out - ref).abs().max()
        again = ops.gemm_nt(A, B, L.EPI_STORE_F32, bias=bias, impl=L.IMPL_MFMA_SK)          # second launch: the flags carry the next epoch
        assert torch.equal(out, again)


@pytest.mark.parametrize("epi", ["gelu", "resid", "dgelu", "store"])
def test_gemm_nt_persistent_kstream_epilogues(epi):
    M, N, K = 264, 264, 256                     # 2 x 2 tiles with 8 valid rows / columns in the edge tiles; 3 workgroups, K cuts
    A, B, bias = rnd(M, K, seed=34).bfloat16(), rnd(N, K, seed=35, scale=K ** -0.5).bfloat16(), rnd(N, seed=36)
    acc = A.float() @ B.float().t()
    with patched() as lib, option(lib, L.OPT_SK, 3), option(lib, L.OPT_SK_GRID, 3):
        if epi == "gelu":
            out, pre = ops.gemm_nt(A, B, L.EPI_GELU, bias=bias, impl=L.IMPL_MFMA_SK)
            assert torch.allclose(pre.float(), (acc + bias).bfloat16().float(), atol=2e-2)
            assert torch.allclose(out.float(), gelu(pre.float()), rtol=2e-2, atol=2e-2)
        elif epi == "resid":
            aux = rnd(M, N, seed=37)
            out = ops.gemm_nt(A, B, L.EPI_RESID_F32, bias=bias, aux=aux, impl=L.IMPL_MFMA_SK)
            assert torch.allclose(out, aux + acc + bias, rtol=1e-4, atol=1e-4)
        elif epi == "dgelu":
            aux = rnd(M, N, seed=38).bfloat16()
            out = ops.gemm_nt(A, B, L.EPI_MUL_DGELU, aux=aux, impl=L.IMPL_MFMA_SK)
            assert torch.allclose(out.float(), acc * dgelu(aux.float()), rtol=3e-2, atol=3e-2)
        else:
            out = ops.gemm_nt(A, B, L.EPI_STORE, bias=bias, impl=L.IMPL_MFMA_SK)
            assert (out.float() - (acc + bias).bfloat16().float()).abs().max() <= 2 * 2 ** -8 * (acc + bias).abs().max()


# ---- round 6: the persistent K-stream weight-gradient kernel (csrc/gemm_tn_sk.hip) ---------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,grid,bias", [
    (256, 256, 256, 1, True),        # one tile, one workgroup, two pairs
    (640, 512, 256, 3, True),        # two tiles on three workgroups (2 + 1): five pairs shared 3 / 2 in the first tile; the bias owners are the tiles of column 0
    (384, 264, 392, 8, False),       # 2 x 2 tiles with ragged N / K (8 and 136 valid columns in the edge tiles), two workgroups per tile, three pairs
    (128, 256, 256, 4, True),        # one pair on four workgroups: three zero slabs
])
def test_gemm_tn_persistent_kstream(M, N, K, grid, bias):
    Y, X = rnd(M, N, seed=41).bfloat16(), rnd(M, K, seed=42).bfloat16()
    dW0 = rnd(N, K, seed=43)
    db0 = rnd(N, seed=44)
    with patched() as lib, option(lib, L.OPT_TN_SK, 1), option(lib, L.OPT_SK_GRID, grid):
        dW, db = dW0.clone(), (db0.clone() if bias else None)
        ops.gemm_tn_wgrad(Y, X, dW=dW, dbias=db)
    ref = dW0.double() + Y.double().t() @ X.double()
    assert torch.allclose(dW.double(), ref, rtol=1e-5, atol=1e-4), (dW.double() - ref).abs().max()
    if bias:
        assert torch.allclose(db.double(), db0.double() + Y.double().sum(0), rtol=1e-5, atol=1e-4)
