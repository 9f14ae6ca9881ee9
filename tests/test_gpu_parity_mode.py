"""The fp32-accurate mode at MFMA speed, at FULL module size (BASELINE.json cfg2 / cfg3 dimensions: C = 512, H = 8, depth 6, 224^2, T = 16; one clip):
compute_dtype = float32, fp32 matrix products on the split-bf16 kernels (process-wide "bf16x3", the backbone on its own "bf16x6" engine -- MAED(
backbone_f32_matmul=...)), every convolution but the 7x7 stem and the two strided 3x3 backward passes on the library's own kernels.

  * outputs vs the fp32 CPU oracle (north_star: 1e-3 relative on SMPL parameters)
  * EVERY parameter gradient of the whole model vs fp64 autograd through the oracle:
      - outside the backbone: 1e-3 of the tensor's largest gradient, per parameter
      - inside the backbone the fp32 REFERENCE ARITHMETIC ITSELF is 1.5e-2 (median over a stage's parameters) / 3.5e-2 (worst) away from fp64 on a freshly
        initialised network -- 52 GroupNorms behind weight-standardised convolutions amplify fp32 rounding (measured here every run: the fp32 oracle vs the
        fp64 oracle).  No fp32 implementation can be closer to fp64 than fp32 arithmetic is; the bar is therefore the reference's own distance: per
        stage, median and worst error at most 1.5x the fp32 oracle's (+1e-3).  bf16x6 meets it (1.0x); bf16x3 in the backbone does not (4x:
        profiles/r03_x3_probe_call2_mixed.txt), which is why the backbone carries its own engine.
"""
import os
import sys

import pytest
import torch

from oracle import maed_ref as R

pytestmark = pytest.mark.gpu

from _util import DEV, note, report, rnd  # noqa: E402

CFG = dict(depth=6, H=8, img=224, hidden=1024, T=16)
WTS = {"theta": 1.0, "kp_3d": 1.0, "kp_2d": 0.01}


def _group(name):
    if "backbone" in name:
        s = name.split("backbone.")[1]
        return "backbone." + (s.split(".")[0] if s.startswith("stem") else ".".join(s.split(".")[:2]))
    return "ste+decoder"


def _oracle(params, clip, sp, dtype):
    pd = {k: v.clone().to(dtype).requires_grad_(True) for k, v in params.items()}
    spd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sp.items()}
    out = R.maed_forward(clip.to(dtype), pd, spd, depth=CFG["depth"], H=CFG["H"])
    sum(w * (out[k] ** 2).mean() for k, w in WTS.items()).backward()
    return {k: v.detach() for k, v in out.items()}, {k: v.grad for k, v in pd.items() if v.grad is not None}


def test_cfg3_full_size_parity_mode_outputs_and_every_gradient():
    import maed_amd
    from maed_amd import ops
    C, P = 64 * CFG["H"], (CFG["img"] // 16) ** 2 + 1
    params = R.make_params(embed_dim=C, depth=CFG["depth"], hidden_dim=CFG["hidden"], n_tokens=P, seed=7)
    sp = R.make_synthetic_smpl(0)
    clip = rnd(1, CFG["T"], 3, CFG["img"], CFG["img"], seed=21)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    o64, g64 = _oracle(params, clip, sp, torch.float64)
    o32, g32 = _oracle(params, clip, sp, torch.float32)
    rel = lambda a, b: float((a.double().cpu() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))

    old = ops.get_float32_matmul_precision()
    calls = {"c1": 0, "c3": 0}
    real_c1, real_c3 = ops.Conv1x1Fn.forward, ops.conv3x3
    try:
        ops.set_float32_matmul_precision("bf16x3")
        ops.Conv1x1Fn.forward = staticmethod(lambda ctx, *a: (calls.__setitem__("c1", calls["c1"] + 1), real_c1(ctx, *a))[1])
        ops.conv3x3 = lambda *a, **k: (calls.__setitem__("c3", calls["c3"] + 1), real_c3(*a, **k))[1]
        m = maed_amd.MAED(num_blocks=CFG["depth"], num_heads=CFG["H"], embed_dim=C, hidden_dim=CFG["hidden"], img_size=CFG["img"], compute_dtype=torch.float32,
                          backbone_f32_matmul="bf16x6")
        m.load_state_dict(params, strict=False)
        m = m.to(DEV).train()
        m.decoder.drop1.p = 0.0
        m.decoder.drop2.p = 0.0
        out = m(clip.to(DEV))
        sum(w * (out[k] ** 2).mean() for k, w in WTS.items()).backward()
        torch.cuda.synchronize()
    finally:
        ops.set_float32_matmul_precision(old)
        ops.Conv1x1Fn.forward, ops.conv3x3 = real_c1, real_c3
    # the library's own convolutions carried the backbone: 35 1x1 (33 stride-1 + 2 packed stride-2 shortcuts), 16 3x3 forward + 14 stride-1 input gradients
    assert calls["c1"] == 35 and calls["c3"] == 16 + 14, calls

    for k in ("theta", "kp_3d", "kp_2d", "rotmat", "verts"):
        report(f"parity mode (f32, bf16x3 / backbone bf16x6) cfg3 full size {k} vs fp32 oracle", out[k].detach().float(), o32[k], rtol=0, atol=1e-3 * o32[k].abs().max().item())
    by_mine, by_ref = {}, {}
    for n, p in m.named_parameters():
        assert p.grad is not None and n in g64, n
        by_mine.setdefault(_group(n), []).append((rel(p.grad, g64[n]), n))
        by_ref.setdefault(_group(n), []).append((rel(g32[n], g64[n]), n))
    for grp in sorted(by_mine):
        a, b = sorted(by_mine[grp]), sorted(by_ref[grp])
        med_a, med_b, w_a, w_b = a[len(a) // 2][0], b[len(b) // 2][0], a[-1][0], b[-1][0]
        note(f"parity mode gradients vs fp64, {grp:20s} n={len(a):3d}: ours median {med_a:.2e} worst {w_a:.2e} ({a[-1][1]});  fp32 oracle median {med_b:.2e} worst {w_b:.2e}")
        if grp == "ste+decoder":
            assert w_a <= 1e-3, (grp, a[-1])
        else:
            assert med_a <= 1.5 * med_b + 1e-3 and w_a <= 1.5 * w_b + 1e-3, (grp, med_a, med_b, w_a, w_b)


def test_cfg3_full_size_mixed_mode_bf16x3_forward_bf16_backward():
    """round 4, "bf16x3 forward / bf16 backward" (maed_amd.set_float32_backward_precision("bf16x1")): fp32 storage, the forward's matrix products split into two bf16
    planes (3 MFMAs), the backward's with ONE plane (MAED_F32X1) -- the fastest mode whose OUTPUTS meet north_star's 1e-3 on SMPL parameters at full module size
    (bench.py parity_mode.fastest_within_1e3).  Asserted: outputs vs the fp32 CPU oracle within 1e-3 of each output's maximum (the same as the plain bf16x3
    forward up to the order of its fp32 atomics: the backward engine must not leak into it), and every parameter gradient against the all-bf16x3 gradients of the same model -- bf16-product level:
    cosine >= 0.995 per tensor outside the backbone, >= 0.95 inside (where fp32 arithmetic itself is 1.5e-2 from fp64: the test above)."""
    import maed_amd
    from maed_amd import ops
    C, P = 64 * CFG["H"], (CFG["img"] // 16) ** 2 + 1
    params = R.make_params(embed_dim=C, depth=CFG["depth"], hidden_dim=CFG["hidden"], n_tokens=P, seed=7)
    sp = R.make_synthetic_smpl(0)
    clip = rnd(1, CFG["T"], 3, CFG["img"], CFG["img"], seed=21)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        o32 = R.maed_forward(clip, params, sp, depth=CFG["depth"], H=CFG["H"])
    old = ops.get_float32_matmul_precision()
    res = {}
    try:
        ops.set_float32_matmul_precision("bf16x3")
        for bwd in (None, "bf16x1", "bf16"):
            ops.set_float32_backward_precision(bwd)
            twins = ops.TWIN_FORWARDS[0]
            m = maed_amd.MAED(num_blocks=CFG["depth"], num_heads=CFG["H"], embed_dim=C, hidden_dim=CFG["hidden"], img_size=CFG["img"], compute_dtype=torch.float32)
            m.load_state_dict(params, strict=False)
            m = m.to(DEV).train()
            m.decoder.drop1.p = 0.0
            m.decoder.drop2.p = 0.0
            out = m(clip.to(DEV))
            sum(w * (out[k] ** 2).mean() for k, w in WTS.items()).backward()
            torch.cuda.synchronize()
            res[bwd] = ({k: v.detach().float().cpu() for k, v in out.items()}, {n: p.grad.detach().float().cpu() for n, p in m.named_parameters()})
            # "bf16" (round 5): the backbone as a bf16 graph over fp32 shadows + every STE block through maed_ste_block_fwd_twin -- and nothing of it in the other modes
            assert ops.TWIN_FORWARDS[0] - twins == ((1 + CFG["depth"]) if bwd == "bf16" else 0), (bwd, ops.TWIN_FORWARDS[0] - twins)
            assert not ops._SHADOW
            del m, out
    finally:
        ops.set_float32_matmul_precision(old)
        ops.set_float32_backward_precision(None)
    for bwd, label in (("bf16x1", "bf16x3 forward / bf16x1 backward"), ("bf16", "bf16x3 forward on fp32 shadows / bf16 backward on bf16 twins")):
        for k in ("theta", "kp_3d", "kp_2d", "rotmat", "verts"):
            report(f"mixed mode (f32 forward storage, {label}) cfg3 full size {k} vs fp32 oracle", res[bwd][0][k], o32[k], rtol=0,
                   atol=1e-3 * o32[k].abs().max().item())
            # same forward arithmetic in all runs; only the order of fp32 atomics (split-K head GEMMs, token means) differs from run to run -- which moves the
            # outputs by up to 1.8e-5 of their maximum between two runs of ONE setting (scripts/x3p_forward_noise.py, profiles/r05_x3p_micro.txt: the first
            # bound of 2e-5 sat inside that noise and tripped at 2.1e-5)
            assert (res[bwd][0][k] - res[None][0][k]).abs().max() <= 5e-5 * o32[k].abs().max(), f"{k}: the backward engine ({bwd}) changed the forward"
        worst = {}
        for n, g1 in res[bwd][1].items():
            g3 = res[None][1][n]
            cos = float((g1.double() * g3.double()).sum() / (g1.double().norm() * g3.double().norm() + 1e-30))
            grp = "backbone" if "backbone" in n else "ste+decoder"
            if grp not in worst or cos < worst[grp][0]:
                worst[grp] = (cos, n)
            assert torch.isfinite(g1).all(), n
        for grp, (cos, n) in sorted(worst.items()):
            note(f"mixed mode ({bwd} backward) gradients vs all-bf16x3 gradients, {grp}: worst cosine {cos:.5f} ({n})")
        # the twin mode's gradients are the bf16 MODE's (bf16-rounded saved activations, bf16 gradient stream): inside the backbone that mode sits at cosine
        # ~0.94 against fp64 on a freshly initialised network; the one-plane mode keeps fp32 saved activations and gradient tensors
        lo = {"bf16x1": (0.995, 0.95), "bf16": (0.99, 0.90)}[bwd]
        assert worst["ste+decoder"][0] >= lo[0] and worst["backbone"][0] >= lo[1], (bwd, worst)


def test_cfg5_full_size_twin_mode_training_pass_vs_oracle_fixture():
    """round 5: the mode bench.py promotes to `value_at_1e3` -- bf16x3 forward on fp32 operands, bf16 backward on bf16 twins -- at the LONG-CLIP configuration
    (BASELINE.json configs[4]: T = 64, 256^2, P = 257, depth 12, dim 768) at full size: the outputs of a TRAINING pass (the pass that takes the twin route: bf16
    autograd graph over fp32 shadows in the backbone, maed_ste_block_fwd_twin in every block; KTD dropout off) against the oracle fixture g15 within 1e-3 of each
    output's maximum, and a finite backward through all of it."""
    import importlib.util
    import numpy as np
    import maed_amd
    from maed_amd import ops
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_golden_cfg5", os.path.join(ROOT, "oracle", "make_golden_cfg5.py"))
    mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
    G = np.load(os.path.join(ROOT, "tests", "golden", "g15_cfg5_full.npz"))
    params, clip = mk.inputs()
    c5 = mk.CFG5
    old = maed_amd.get_float32_matmul_precision()
    try:
        maed_amd.set_float32_matmul_precision("bf16x3")
        maed_amd.set_float32_backward_precision("bf16")
        m = maed_amd.MAED(num_blocks=c5["depth"], num_heads=c5["H"], embed_dim=c5["C"], hidden_dim=c5["hidden"], img_size=c5["img"], max_seqlen=c5["T"],
                          compute_dtype=torch.float32)
        m.load_state_dict(params, strict=False)
        m = m.to(DEV).train()
        m.decoder.drop1.p = 0.0
        m.decoder.drop2.p = 0.0
        twins = ops.TWIN_FORWARDS[0]
        out = m(clip.to(DEV))
        assert ops.TWIN_FORWARDS[0] - twins == 1 + c5["depth"]
        sum(w * (out[k] ** 2).mean() for k, w in WTS.items()).backward()
        torch.cuda.synchronize()
        bad = [n for n, p in m.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
        assert not [n for n in bad if "smpl" not in n], bad
    finally:
        maed_amd.set_float32_matmul_precision(old)
        maed_amd.set_float32_backward_precision(None)
    out = dict({k: v.detach() for k, v in out.items()}, verts_sample=out["verts"].detach()[:, :, ::53])
    for k in ("theta", "kp_3d", "kp_2d", "rotmat", "verts_sample"):
        ref = torch.from_numpy(G[k])
        report(f"cfg5 full size, twin mode training pass {k} vs oracle fixture", out[k], ref, rtol=0, atol=1e-3 * ref.abs().max().item())


def test_cfg3_twin_mode_train_steps_track_the_all_bf16x3_mode():
    """What a trainer cares about: the accurate mode's loss trajectory with the bf16 backward on twins.  Six FusedAdam steps on one fixed 2-clip batch at cfg3
    dimensions with the reference's own objective (LossVideo), from the same seeded parameters, KTD dropout off: all-bf16x3 (forward and backward on split products)
    against the twin mode.  The first loss is the same forward (to the order of its fp32 atomics); the later ones follow it although every update came from bf16-level
    gradients -- within 1e-3 of each other's value, both decreasing."""
    import maed_amd
    from maed_amd import ops
    from maed_amd.ddp import FusedAdam, GradBucketer, ParamArena
    from maed_amd.loss import LossVideo
    C, P = 64 * CFG["H"], (CFG["img"] // 16) ** 2 + 1
    params = R.make_params(embed_dim=C, depth=CFG["depth"], hidden_dim=CFG["hidden"], n_tokens=P, seed=7)
    g = torch.Generator().manual_seed(5)
    N, T = 2, CFG["T"]
    clip = torch.randn(N, T, 3, CFG["img"], CFG["img"], generator=g).to(DEV)
    r = lambda *s: torch.randn(*s, generator=g)
    tgt = {k: v.to(DEV) for k, v in dict(kp_2d=torch.cat([r(N, T, 49, 2) * 0.3, torch.rand(N, T, 49, 1, generator=g)], -1),
                                         kp_3d=torch.cat([r(N, T, 49, 3) * 0.3, torch.ones(N, T, 49, 1)], -1),
                                         theta=torch.cat([r(N, T, 3) * 0.1, r(N, T, 72) * 0.2, r(N, T, 10)], -1), w_smpl=torch.ones(N, T)).items()}
    crit = LossVideo(e_loss_weight=300.0, e_3d_loss_weight=600.0, e_pose_loss_weight=60.0, e_shape_loss_weight=0.06, e_smpl_norm_loss=1.0, e_smpl_accl_loss=0.0)
    old = ops.get_float32_matmul_precision()
    curves = {}
    try:
        ops.set_float32_matmul_precision("bf16x3")
        for bwd in (None, "bf16"):
            ops.set_float32_backward_precision(bwd)
            m = maed_amd.MAED(num_blocks=CFG["depth"], num_heads=CFG["H"], embed_dim=C, hidden_dim=CFG["hidden"], img_size=CFG["img"], compute_dtype=torch.float32)
            m.load_state_dict(params, strict=False)
            m = m.to(DEV).train()
            m.decoder.drop1.p = m.decoder.drop2.p = 0.0
            arena = ParamArena(m)
            opt = FusedAdam(arena, lr=1e-4, bucketer=GradBucketer(arena, m))
            twins = ops.TWIN_FORWARDS[0]
            losses = []
            for _ in range(6):
                opt.zero_grad()
                loss, _ = crit(m(clip), tgt, None)
                loss.backward()
                opt.step()
                losses.append(loss.detach())
            torch.cuda.synchronize()
            assert ops.TWIN_FORWARDS[0] - twins == (6 * (1 + CFG["depth"]) if bwd == "bf16" else 0)
            curves[bwd] = torch.stack(losses).float().cpu()
            del m, arena, opt
    finally:
        ops.set_float32_matmul_precision(old)
        ops.set_float32_backward_precision(None)
    a, b = curves[None], curves["bf16"]
    note(f"six train steps at cfg3 dims, LossVideo: all-bf16x3 {[round(float(v), 5) for v in a]}  twin mode {[round(float(v), 5) for v in b]}")
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    assert abs(float(a[0] - b[0])) <= 1e-5 * abs(float(a[0])), (a[0], b[0])
    assert ((a - b).abs() <= 1e-3 * a.abs()).all(), (a, b)
    assert a[-1] < a[0] and b[-1] < b[0]
