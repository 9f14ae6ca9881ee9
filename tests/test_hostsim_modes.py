"""The st_modes other than 'parallel' (maed_amd/ste_modes.py; reference lib/models/vision_transformer.py:136-178) with the
kernels running on the host simulator: against the fixture the reference's own Attention / Block / VisionTransformer produced
(tests/golden/g13_st_modes.npz, f32 parity mode) and against fp64 autograd through the CPU oracle (bf16 mode)."""
import os
from functools import partial

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import maed_ref as R
from maed_amd.resnetv2 import ResNetV2
from maed_amd.vision_transformer import Attention, Block, VisionTransformer

from _hostsim import patched
from _util import rnd

MODES = ["series", "vanilla", "temporal", "coupling"]
LN = partial(nn.LayerNorm, eps=1e-6)


def t(a):
    return torch.from_numpy(np.asarray(a))


def sd(fx, prefix="sd.", drop=("ts_attn",)):
    return {k[len(prefix):]: t(fx[k]) for k in fx.files if k.startswith(prefix) and not any(d in k for d in drop)}


def rel(a, b):
    b = b.detach() if torch.is_tensor(b) else torch.from_numpy(np.asarray(b))
    a, b = a.detach().double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def check_param_grads(module, fx, prefix, tol):
    step = int(fx["row_step"])
    for n, p in module.named_parameters():
        want = fx[prefix + n]
        assert p.grad is not None, n
        got = p.grad[::step] if p.dim() == 2 else p.grad
        assert rel(got, want) < tol, (n, rel(got, want))


@pytest.mark.parametrize("mode", MODES)
def test_attention_modes_match_reference(golden, mode):
    fx, g1 = golden("g13_st_modes"), golden("g1_attention")
    H, T = int(g1["heads"]), int(g1["seqlen"])
    att = Attention(128, num_heads=H, qkv_bias=True, st_mode=mode)
    assert not hasattr(att, "ts_attn")
    att.load_state_dict(sd(g1))                                   # strict: same keys as the reference in this mode
    x = t(g1["x"]).clone().requires_grad_(True)
    with patched():
        out = att(x, T, compute_dtype=torch.float32)
        assert out.shape == fx[f"{mode}.att.out"].shape           # (F,1,C) in 'temporal' mode
        (out * t(fx["cot_tok"])[:, :out.shape[1]]).sum().backward()
    assert rel(out, fx[f"{mode}.att.out"]) < 2e-5
    assert rel(x.grad, fx[f"{mode}.att.dx"]) < 1e-4
    check_param_grads(att, fx, f"{mode}.att.grad.", 2e-4)


@pytest.mark.parametrize("mode", MODES)
def test_block_modes_match_reference(golden, mode):
    fx, g2 = golden("g13_st_modes"), golden("g2_block")
    H, T = int(g2["heads"]), int(g2["seqlen"])
    blk = Block(128, H, mlp_ratio=4, qkv_bias=True, norm_layer=LN, st_mode=mode, compute_dtype=torch.float32)
    blk.load_state_dict(sd(g2))
    assert blk.fused_parameters() == []                           # autograd owns every gradient in the staged modes
    x = t(g2["x"]).clone().requires_grad_(True)
    with patched():
        out = blk(x, T)
        (out * t(fx["cot_tok"])).sum().backward()
    assert rel(out, fx[f"{mode}.blk.out"]) < 2e-5
    assert rel(x.grad, fx[f"{mode}.blk.dx"]) < 1e-4
    check_param_grads(blk, fx, f"{mode}.blk.grad.", 2e-4)


@pytest.mark.parametrize("mode", MODES)
def test_block_modes_bf16_vs_fp64_oracle(mode):
    N, T, P, H = 2, 2, 9, 2
    C, Fr = 64 * H, N * T
    p = {k[len("encoder.blocks.0."):]: v for k, v in R.make_params(embed_dim=C, depth=1, hidden_dim=64, layers=(1, 1, 1), n_tokens=P, seed=3).items()
         if k.startswith("encoder.blocks.0.") and "ts_attn" not in k}
    p = {k: v * (3.0 if k.endswith("weight") and v.dim() == 2 else 1.0) for k, v in p.items()}
    x, dy = rnd(Fr, P, C, seed=1), rnd(Fr, P, C, seed=2)
    pd = {k: v.double().requires_grad_(True) for k, v in p.items()}
    xr = x.double().requires_grad_(True)
    yref = R.block(xr, pd, "", H, T, mode)
    yref.backward(dy.double())
    blk = Block(C, H, mlp_ratio=4, qkv_bias=True, norm_layer=LN, st_mode=mode, compute_dtype=torch.bfloat16)
    blk.load_state_dict(p)
    xg = x.clone().requires_grad_(True)
    with patched():
        y = blk(xg, T)
        y.backward(dy)
    assert rel(y, yref) < 3e-2
    assert rel(xg.grad, xr.grad) < 3e-2
    for name, prm in blk.named_parameters():
        assert prm.grad is not None, name
        assert rel(prm.grad, pd[name].grad) < 6e-2, (name, rel(prm.grad, pd[name].grad))


@pytest.mark.parametrize("mode", ["series", "temporal"])      # one mode with temp_embed, one without (oracle test covers all 4)
def test_vit_tiny_modes_match_reference(golden, mode):
    fx, g4 = golden("g13_st_modes"), golden("g4_vit_tiny")
    layers = tuple(int(v) for v in g4["layers"])
    bb = ResNetV2(layers=layers, channels=(128, 256, 512), in_chans=3, compute_dtype=torch.float32)
    vit = VisionTransformer(img_size=32, patch_size=16, embed_dim=128, depth=int(g4["depth"]), num_heads=int(g4["heads"]), hybrid_backbone=bb,
                            mlp_ratio=4, qkv_bias=True, representation_size=128, norm_layer=LN, st_mode=mode, num_classes=-1,
                            compute_dtype=torch.float32)
    has_temp = bool(fx[f"{mode}.vit.has_temp_embed"])
    assert hasattr(vit, "temp_embed") == has_temp and ("temp_embed" in vit.state_dict()) == has_temp
    vit.load_state_dict(sd(g4, drop=("ts_attn",) if has_temp else ("ts_attn", "temp_embed")))      # strict
    vit.eval()
    img = t(g4["img"])
    with patched(module_paths=False):                     # backbone on ATen: this test is about the encoder modes
        out = vit(img, seqlen=int(g4["seqlen"]))
        (out * t(fx["cot_feat"])).sum().backward()
    assert rel(out, fx[f"{mode}.vit.out"]) < 1e-4
    pr = dict(vit.named_parameters())
    for k in fx.files:
        if k.startswith(f"{mode}.vit.grad."):
            n = k[len(f"{mode}.vit.grad."):]
            assert rel(pr[n].grad, fx[k]) < 5e-4, (n, rel(pr[n].grad, fx[k]))


def test_unknown_mode_rejected():
    with pytest.raises(NotImplementedError):
        Attention(128, num_heads=2, st_mode="bogus")


def test_staged_block_trains_through_arena_bucketer_and_fused_adam():
    """a staged Block has no kernel-written gradients (fused_parameters() == []): every parameter must reach the gradient arena
    through autograd's accumulate hooks, complete its bucket, and take the same Adam step as torch.optim.Adam"""
    import copy
    from maed_amd.ddp import FusedAdam, GradBucketer, ParamArena
    torch.manual_seed(0)
    blk = Block(128, 2, mlp_ratio=2, qkv_bias=True, norm_layer=LN, st_mode="series", compute_dtype=torch.float32)
    ref = copy.deepcopy(blk)
    x, dy = rnd(4, 5, 128, seed=3), rnd(4, 5, 128, seed=4)
    with patched():
        arena = ParamArena(blk, device=torch.device("cpu"))
        bucketer = GradBucketer(arena, blk, bucket_bytes=64 << 10)
        assert len(bucketer.buckets) > 1 and not bucketer._fused
        opt = FusedAdam(arena, lr=1e-2, weight_decay=1e-3, bucketer=bucketer, model=blk)
        topt = torch.optim.Adam(ref.parameters(), lr=1e-2, weight_decay=1e-3)
        for step in range(2):
            opt.zero_grad()
            (blk(x, 2) * dy).sum().backward()
            assert all(bucketer._launched), "every bucket must have completed during backward"
            topt.zero_grad()
            (ref(x, 2) * dy).sum().backward()
            for (n, p), q_ in zip(blk.named_parameters(), ref.parameters()):
                assert p.grad.data_ptr() == arena.grad[arena.offsets[arena.index[id(p)]]:].data_ptr(), n      # accumulated IN the arena
                if step == 0:
                    torch.testing.assert_close(p.grad, q_.grad, rtol=1e-4, atol=1e-5, msg=n)
                # (softmax is invariant to the key bias: that gradient is pure rounding noise, which Adam's 1/sqrt(v) would turn
                #  into O(lr) differences) -> both optimizers step on the SAME gradients
                q_.grad.copy_(p.grad)
            opt.step()
            topt.step()
    for (n, p), q_ in zip(blk.named_parameters(), ref.parameters()):
        torch.testing.assert_close(p, q_, rtol=1e-5, atol=1e-6, msg=n)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4),
                                       pytest.param(torch.bfloat16, 6e-2, marks=pytest.mark.skipif(os.environ.get("MAED_SLOW_TESTS") != "1",
                                                                                                   reason="40 s on the simulator: MAED_SLOW_TESTS=1"))])
def test_vit_tiny_parallel_mode_whole_gpu_path_matches_reference(golden, dtype, tol):
    """the benchmarked configuration (st_mode='parallel', fused STE blocks) with the BACKBONE also on its library path (batched weight
    standardisation, fused GroupNorm, max-pool kernels; GEMM convolutions in bf16), against the reference's own output (g4)"""
    g4 = golden("g4_vit_tiny")
    layers = tuple(int(v) for v in g4["layers"])
    bb = ResNetV2(layers=layers, channels=(128, 256, 512), in_chans=3, compute_dtype=dtype)
    vit = VisionTransformer(img_size=32, patch_size=16, embed_dim=128, depth=int(g4["depth"]), num_heads=int(g4["heads"]), hybrid_backbone=bb,
                            mlp_ratio=4, qkv_bias=True, representation_size=128, norm_layer=LN, st_mode="parallel", num_classes=-1,
                            compute_dtype=dtype)
    vit.load_state_dict(sd(g4, drop=()))                              # strict: every key of the reference, ts_attn and temp_embed included
    vit.eval()
    seen = {}
    bb.register_forward_hook(lambda m, i, o: seen.__setitem__("feat", o))
    with patched(), torch.no_grad():
        out = vit(t(g4["img"]), seqlen=int(g4["seqlen"]))              # on_library_device -> the HIP backbone path, not ATen
    assert seen["feat"].dtype == dtype
    assert rel(seen["feat"].float(), g4["backbone_out"]) < (1e-4 if dtype == torch.float32 else 5e-2)
    assert rel(out, g4["out"]) < tol
