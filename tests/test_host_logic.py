"""CPU tests of the host side: module API / state_dict names against the reference, the ATen parts of
the package (backbone, decoder training tail) against the golden fixtures, and the data-parallel
runtime (flat arenas, bucketed all-reduce) on 2 gloo ranks."""
import os
from functools import partial

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import maed_ref as R


def t(a):
    return torch.from_numpy(np.asarray(a))


def sd(fx, prefix):
    return {k[len(prefix):]: t(fx[k]) for k in fx.files if k.startswith(prefix)}


def test_state_dict_names_match_reference(golden):
    import maed_amd
    fx = golden("g10_cfg1_full")
    m = maed_amd.MAED()  # reference defaults: ste / 6 blocks / 12 heads / parallel / ktd / 1024
    ours = {k for k in m.state_dict() if ".smpl." not in k}
    ref = {k for k in fx["state_dict_keys"].tolist() if ".smpl." not in k}
    assert ours == ref
    assert sum(p.numel() for p in m.parameters()) == int(fx["n_params"])
    assert m.encoder_type == "ste" and m.decoder_type == "ktd"
    # DDP checkpoints carry a 'module.' prefix; smpl buffers are dropped on load (eval.py:29)
    params = R.make_params(seed=1)
    missing, unexpected = m.load_state_dict(params, strict=False)
    assert not unexpected and all(".smpl." in k for k in missing)


def test_unsupported_variants_raise_like_reference():
    import maed_amd
    with pytest.raises(NotImplementedError):
        maed_amd.MAED(encoder="foo")
    with pytest.raises(NotImplementedError):
        maed_amd.MAED(decoder="foo")
    with pytest.raises(NotImplementedError):
        maed_amd.Attention(128, 2, st_mode="bogus")


@pytest.mark.parametrize("tag", ["odd", "even"])
def test_backbone_pieces_golden(golden, tag):
    from maed_amd import resnetv2 as rn
    fx = golden("g5_backbone_pieces")
    x = t(fx[f"{tag}.x"])
    c3 = rn.StdConv2dSame(32, 64, 3, stride=2)
    c3.weight.data = t(fx[f"{tag}.w3"])
    c7 = rn.StdConv2dSame(32, 64, 7, stride=2)
    c7.weight.data = t(fx[f"{tag}.w7"])
    gn = rn.GroupNormAct(64)
    gn.weight.data, gn.bias.data = t(fx[f"{tag}.gn.weight"]), t(fx[f"{tag}.gn.bias"])
    with torch.no_grad():
        y3 = c3(x)
        np.testing.assert_allclose(y3.numpy(), fx[f"{tag}.conv3s2"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(c7(x).numpy(), fx[f"{tag}.conv7s2"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(gn(y3).numpy(), fx[f"{tag}.gn_relu"], rtol=1e-4, atol=1e-4)
        np.testing.assert_array_equal(rn.MaxPool2dSame(3, 2)(x).numpy(), fx[f"{tag}.maxpool"])
        bt = rn.Bottleneck(32, 128, stride=2, downsample=True)
        bt.load_state_dict(sd(fx, f"{tag}.bt."))
        np.testing.assert_allclose(bt(x).numpy(), fx[f"{tag}.bottleneck"], rtol=1e-4, atol=1e-4)


def test_backbone_tiny_golden(golden):
    from maed_amd.resnetv2 import ResNetV2
    fx = golden("g4_vit_tiny")
    bb = ResNetV2(layers=(1, 1, 1), channels=(128, 256, 512))
    bb.load_state_dict(sd(fx, "sd.patch_embed.backbone."))
    with torch.no_grad():
        np.testing.assert_allclose(bb(t(fx["img"])).numpy(), fx["backbone_out"], rtol=1e-4, atol=1e-4)


def test_ktd_training_tail_golden(golden):
    """the ATen (training) path of the decoder against the reference's outputs (eval mode: dropout off)"""
    from maed_amd.ktd import KTD
    fx = golden("g6_ktd")
    dec = KTD(feat_dim=128, hidden_dim=64).eval()
    dec.load_state_dict(sd(fx, "sd."), strict=False)
    with torch.no_grad():
        pose, shape, cam = dec._head_torch(t(fx["x"]))
        o = dec(t(fx["x"]), seqlen=3)
        o17 = dec(t(fx["x"]), seqlen=3, J_regressor=R.make_synthetic_smpl(0)["J_regressor_h36m"])
    np.testing.assert_allclose(pose.numpy(), fx["pose6d"], rtol=2e-5, atol=2e-5)
    for k in ("theta", "rotmat", "kp_3d"):
        np.testing.assert_allclose(o[k].numpy(), fx[k], rtol=1e-4, atol=2e-5, err_msg=k)
    np.testing.assert_allclose(o["kp_2d"].numpy(), fx["kp_2d"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(o["verts"][:, ::53].numpy(), fx["verts_sub"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(o17["kp_3d"].numpy(), fx["kp_3d_h36m"], rtol=1e-4, atol=2e-5)
    assert np.array_equal(dec.smpl.joint_map.numpy(), fx["joint_map"])  # integer table: bit-exact


def test_geometry_golden(golden):
    from maed_amd import geometry, spin
    fx = golden("g7_geometry")
    np.testing.assert_allclose(geometry.rot6d_to_rotmat(t(fx["rot6d"])).numpy(), fx["rotmat"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(geometry.rotation_matrix_to_angle_axis(t(fx["rotmat_all"])).numpy(), fx["angle_axis"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(spin.projection(t(fx["joints"]), t(fx["cam"])).numpy(), fx["kp_2d"], rtol=1e-5, atol=1e-5)


def test_synthetic_smpl_matches_oracle_stand_in():
    from maed_amd.smpl import SMPL, synthetic_smpl_arrays
    a, b = synthetic_smpl_arrays(0), R.make_synthetic_smpl(0)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    s = SMPL()
    assert s.joint_map.tolist() == R.JOINT_MAP_49
    assert s.parents.tolist() == R.SMPL_PARENTS


# ---------------------------------------------------------------------------------------------------
# data-parallel runtime on 2 gloo ranks (CPU)
# ---------------------------------------------------------------------------------------------------
class _Toy(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = nn.Linear(16, 32)
        self.b = nn.Linear(32, 32)
        self.c = nn.Linear(32, 4)

    def forward(self, x):
        return self.c(torch.tanh(self.b(torch.tanh(self.a(x)))))


def _ddp_worker(rank, world, port, out):
    import torch.distributed as dist
    from maed_amd.ddp import GradBucketer, ParamArena
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _Toy()
        if rank == 1:  # rank 1 starts from different weights: broadcast_parameters must fix that
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(1.0)
        arena = ParamArena(model, device=torch.device("cpu"))
        bucketer = GradBucketer(arena, model, bucket_bytes=2048)  # several small buckets
        bucketer.broadcast_parameters(0)
        assert len(bucketer.buckets) > 1
        x = torch.randn(8, 16, generator=torch.Generator().manual_seed(42))
        y = torch.randn(8, 4, generator=torch.Generator().manual_seed(43))
        shard = slice(rank * 4, rank * 4 + 4)
        for _ in range(2):  # two steps: counters must reset
            arena.zero_grad()
            # two forwards, one backward (lib/core/trainer.py:188-202)
            loss = ((model(x[shard]) - y[shard]) ** 2).mean() + 0.5 * ((model(x[shard] * 2) - y[shard]) ** 2).mean()
            loss.backward()
            bucketer.finish()
        out[rank] = (arena.grad / world).clone(), arena.flat.clone()
    finally:
        dist.destroy_process_group()


def test_gradient_bucketing_allreduce_gloo_world2():
    import torch.multiprocessing as mp
    world, port = 2, 29533 + os.getpid() % 1000
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_ddp_worker, args=(world, port, out), nprocs=world, join=True)
    g0, p0 = out[0]
    g1, p1 = out[1]
    assert torch.equal(p0, p1), "parameters must be identical after broadcast"
    assert torch.equal(g0, g1), "every rank must hold the same reduced gradient"
    # single-process reference on the concatenated batch
    from maed_amd.ddp import ParamArena
    model = _Toy()
    arena = ParamArena(model, device=torch.device("cpu"))
    x = torch.randn(8, 16, generator=torch.Generator().manual_seed(42))
    y = torch.randn(8, 4, generator=torch.Generator().manual_seed(43))
    loss = ((model(x) - y) ** 2).mean() + 0.5 * ((model(x * 2) - y) ** 2).mean()
    loss.backward()
    torch.testing.assert_close(g0, arena.grad, rtol=1e-5, atol=1e-6)


def test_param_arena_views_and_order():
    import maed_amd
    from maed_amd.ddp import GradBucketer, ParamArena
    m = maed_amd.MAED(num_blocks=2, num_heads=2, embed_dim=128, hidden_dim=64, img_size=32, compute_dtype=torch.float32)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    arena = ParamArena(m, device=torch.device("cpu"))
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k
    for p, o in zip(arena.params, arena.offsets):
        assert p.data_ptr() == arena.flat.data_ptr() + 4 * o and o % 64 == 0
        assert p.grad.data_ptr() == arena.grad.data_ptr() + 4 * o
    groups = [0 if n.startswith("encoder.patch_embed.backbone") else 1 if n.startswith("encoder.patch_embed") else
              3 if n.startswith("encoder.blocks") else 5 if n.startswith("decoder") else 2 for n in arena.names]
    first_block, last_backbone = groups.index(3), max(i for i, g in enumerate(groups) if g == 0)
    assert last_backbone < first_block < groups.index(5)
    b = GradBucketer(arena, m, bucket_bytes=1 << 20)
    covered = sorted((s, e) for s, e, _ in b.buckets)
    assert covered[0][0] == 0 and covered[-1][1] == arena.numel
    assert all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
    assert sum(n for _, _, n in b.buckets) == len(arena.params)


class _StubMAED(torch.nn.Module):
    """CPU stand-in with MAED's output contract (clip (N,T,3,H,W) -> dict of (N,T,...) tensors): the train-step logic under test is
    model-agnostic, and the real STE has no CPU path by design"""

    def __init__(self):
        super().__init__()
        self.enc = torch.nn.Linear(3, 16)
        self.dec = torch.nn.Linear(16, 49 * 2 + 49 * 3 + 85)
        self.register_buffer("smpl_like", torch.zeros(3))

    def forward(self, x, J_regressor=None):
        N, T = x.shape[:2]
        o = self.dec(torch.tanh(self.enc(x.mean(dim=(-1, -2)))))
        return dict(kp_2d=o[..., :98].reshape(N, T, 49, 2), kp_3d=o[..., 98:245].reshape(N, T, 49, 3), theta=o[..., 245:])


def test_train_step_semantics_and_checkpoint_format(tmp_path):
    """maed_amd.trainer.TrainStep == lib/core/trainer.py:159-204,240-262 (video forward over cat(2D clips, 3D clips), image
    forward with T = 1, frame-count weights, one backward, merge_loss), and checkpoints in the reference's layout
    (trainer.py:330-368): 'module.'-prefixed state_dict + torch.optim.Adam-style optimizer state."""
    import copy
    from maed_amd.loss import Loss
    from maed_amd.trainer import TrainStep, load_checkpoint, save_checkpoint
    torch.manual_seed(0)
    model = _StubMAED()
    crit = Loss(device="cpu")
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g)
    T = 2
    t2d = dict(images=r(1, T, 3, 8, 8), kp_2d=torch.cat([r(1, T, 49, 2), torch.rand(1, T, 49, 1, generator=g)], -1))
    t3d = dict(images=r(2, T, 3, 8, 8), kp_2d=torch.cat([r(2, T, 49, 2), torch.ones(2, T, 49, 1)], -1),
               kp_3d=torch.cat([r(2, T, 49, 3), torch.ones(2, T, 49, 1)], -1), theta=r(2, T, 85) * 0.2, w_smpl=torch.tensor([[1., 0.], [1., 1.]]))
    timg = dict(image=r(3, 3, 8, 8), kp_2d=torch.cat([r(3, 49, 2), torch.ones(3, 49, 1)], -1), theta=r(3, 85) * 0.2, w_smpl=torch.ones(3),
                kp_3d=torch.cat([r(3, 49, 3), torch.ones(3, 49, 1)], -1))
    # expected, spelled out as the reference does it
    ref = copy.deepcopy(model)
    lv, dv = crit(preds=ref(torch.cat((t2d["images"], t3d["images"]), 0)), target_3d=t3d, target_2d=t2d)
    li, di = crit(preds=ref(timg["image"].unsqueeze(1)), target_img=timg)
    nt_vid, nt_img = 3 * T, 3
    w_vid = nt_vid / (nt_img + nt_vid)
    (li * (1 - w_vid) + lv * w_vid).backward()
    exp_total, exp_terms = crit.merge_loss(lv, dv, li, di, vid_w=w_vid, img_w=1 - w_vid)
    opt = torch.optim.Adam([{"params": p, "name": n} for n, p in model.named_parameters()], lr=1e-3)    # utils.py:127-132
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    total, terms = TrainStep(model, crit, opt)(target_2d=t2d, target_3d=t3d, target_img=timg)
    assert torch.allclose(total, exp_total) and terms.keys() == exp_terms.keys()
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-8), n
    assert all(not torch.equal(before[n], p) for n, p in model.named_parameters()), "the optimizer stepped"
    # video-only and image-only iterations (img_use_freq > 1: trainer.py:148)
    tot_v, _ = TrainStep(model, crit, opt)(target_3d=t3d)
    tot_i, _ = TrainStep(model, crit, opt)(target_img=timg)
    assert torch.isfinite(tot_v) and torch.isfinite(tot_i)
    # checkpoint layout
    path = str(tmp_path / "epoch_1.pth.tar")
    save_checkpoint(path, model, opt, epoch=1, performance=55.5)
    ck = torch.load(path, map_location="cpu")
    assert set(ck) == {"epoch", "state_dict", "performance", "optimizer"} and all(k.startswith("module.") for k in ck["state_dict"])
    assert [gr["name"] for gr in ck["optimizer"]["param_groups"]] == [n for n, _ in model.named_parameters()]
    fresh = _StubMAED()
    fopt = torch.optim.Adam([{"params": p, "name": n} for n, p in fresh.named_parameters()], lr=1e-3)
    epoch, perf = load_checkpoint(path, fresh, fopt)
    assert (epoch, perf) == (1, 55.5)
    for (n, p), (_, q) in zip(fresh.named_parameters(), model.named_parameters()):
        assert torch.equal(p, q), n
    assert len(fopt.state_dict()["state"]) == len(opt.state_dict()["state"])


class _ToyFused(nn.Module):
    """autograd-managed layers + a module whose kernels write .grad directly and report through `grads_ready` (the KTD
    regressor head on the host simulator) -- the two kinds of parameters the gradient bucketer has to track"""

    def __init__(self):
        super().__init__()
        from maed_amd.ktd import KTD
        torch.manual_seed(0)
        self.enc = nn.Linear(12, 48)
        self.decoder = KTD(feat_dim=48, hidden_dim=32)
        self.decoder.drop1.p = self.decoder.drop2.p = 0.0
        for r in self.decoder._regressors():
            nn.init.normal_(r.weight, std=0.05)

    def forward(self, x):
        pose, shape, cam = self.decoder._head_train(torch.tanh(self.enc(x)))
        return torch.cat([pose, shape, cam], 1)


def _ddp_step_worker(rank, world, port, out):
    import torch.distributed as dist
    from maed_amd.ddp import FusedAdam, GradBucketer, ParamArena
    from _hostsim import patched
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with patched():
            model = _ToyFused()
            arena = ParamArena(model, device=torch.device("cpu"))
            bucketer = GradBucketer(arena, model, bucket_bytes=4096)
            bucketer.broadcast_parameters(0)
            opt = FusedAdam(arena, lr=1e-2, weight_decay=1e-3, bucketer=bucketer, model=model)
            assert len(bucketer.buckets) > 2 and len(bucketer._fused) == 52
            x = torch.randn(8, 12, generator=torch.Generator().manual_seed(7))
            shard = slice(rank * 4, rank * 4 + 4)
            for _ in range(3):
                opt.zero_grad()
                (model(x[shard]) ** 2).mean().backward()
                opt.step()
            out[rank] = arena.flat.clone()
    finally:
        dist.destroy_process_group()


def test_full_ddp_train_steps_gloo_world2_match_single_process_adam():
    """whole data-parallel steps on 2 gloo ranks (bucketed all-reduce incl. a fused-gradient module, 1/world folded into the Adam
    kernel -- run on the host simulator) == torch.optim.Adam on the concatenated batch in one process"""
    import torch.multiprocessing as mp
    world, port = 2, 30533 + os.getpid() % 1000
    out = mp.Manager().dict()
    mp.spawn(_ddp_step_worker, args=(world, port, out), nprocs=world, join=True)
    assert torch.equal(out[0], out[1]), "ranks must stay bit-identical"
    from maed_amd.ddp import ParamArena
    from _hostsim import patched
    with patched():
        ref = _ToyFused()
        arena = ParamArena(ref, device=torch.device("cpu"))
        opt = torch.optim.Adam(ref.parameters(), lr=1e-2, weight_decay=1e-3)
        x = torch.randn(8, 12, generator=torch.Generator().manual_seed(7))
        for _ in range(3):
            opt.zero_grad(set_to_none=False)
            arena.zero_grad()
            (ref(x) ** 2).mean().backward()
            opt.step()
    # Adam divides by sqrt(v): an element whose gradient is ~0 amplifies summation-order noise -> absolute tolerance of 1% of one lr step
    torch.testing.assert_close(out[0], arena.flat, rtol=1e-3, atol=1e-4)


def test_smpl_derived_tables_follow_a_loaded_state_dict():
    """the reference's Trainer.resume_pretrained loads checkpoints that carry decoder.smpl.* with strict=True
    (lib/core/trainer.py:358); the joint tables folded from J_regressor / v_template / shapedirs and the backward's GEMM operand
    must follow the loaded arrays, not stay on the construction-time (synthetic) ones."""
    from maed_amd.smpl import SMPL, synthetic_smpl_arrays
    a, b = SMPL(synthetic_smpl_arrays(0)), SMPL(synthetic_smpl_arrays(7))
    assert not torch.allclose(a.J_template, b.J_template)
    ps_old = a.pose_shape_dirs_t().clone()
    a.load_state_dict(b.state_dict(), strict=True)
    assert torch.equal(a.J_template, b.J_template) and torch.equal(a.J_shapedirs, b.J_shapedirs)
    assert torch.equal(a.pose_shape_dirs_t(), b.pose_shape_dirs_t()) and not torch.equal(a.pose_shape_dirs_t(), ps_old)
    g = torch.Generator().manual_seed(1)
    betas, rot = torch.randn(2, 10, generator=g), torch.eye(3).expand(2, 24, 3, 3).contiguous()
    va, _ = a.lbs_torch(betas, rot)
    vb, _ = b.lbs_torch(betas, rot)
    assert torch.equal(va, vb)
    # in-place edits of a base array are picked up lazily as well
    with torch.no_grad():
        a.v_template.mul_(2.0)
    a.refresh_derived()
    assert torch.allclose(a.J_template, 2.0 * b.J_template, rtol=1e-5, atol=1e-6)
    # the stand-in announces itself unless a test / benchmark opted out
    import os, warnings
    old = os.environ.pop("MAED_SYNTHETIC_SMPL_OK", None)
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            s = SMPL()
        assert s.synthetic and any("synthetic stand-in" in str(x.message) for x in w)
    finally:
        if old is not None:
            os.environ["MAED_SYNTHETIC_SMPL_OK"] = old


def test_weight_initialisation_distributions_follow_the_reference():
    """trunc_normal(std 0.02) for Linear weights and the embeddings with zero biases and unit LayerNorm (vision_transformer.py:357-375), kaiming
    normal fan_out for the backbone's convolutions (resnetv2.py:330-335), xavier uniform with gain 0.01 for the KTD regressors (ktd.py:56-61) -- checked on the
    empirical moments / ranges of a freshly built model"""
    import math
    import maed_amd
    os.environ.setdefault("MAED_SYNTHETIC_SMPL_OK", "1")
    torch.manual_seed(0)
    m = maed_amd.MAED(num_blocks=2, num_heads=8, embed_dim=512, hidden_dim=1024)
    enc, dec = m.encoder, m.decoder
    for name, mod in enc.blocks.named_modules():
        if isinstance(mod, torch.nn.Linear):
            w = mod.weight.detach()
            assert abs(w.std().item() - 0.02) < 0.002 and abs(w.mean().item()) < 1e-3 and w.abs().max().item() <= 2.0, name      # truncation at +-2 (absolute), as the reference
            assert mod.bias is None or bool((mod.bias == 0).all())
        if isinstance(mod, torch.nn.LayerNorm):
            assert bool((mod.weight == 1).all()) and bool((mod.bias == 0).all())
    for t in (enc.pos_embed, enc.temp_embed):
        assert abs(t.std().item() - 0.02) < 0.003
    convs = [c for c in enc.patch_embed.backbone.modules() if isinstance(c, torch.nn.Conv2d) and c.weight.numel() > 50000]
    for c in convs[:8]:
        fan_out = c.out_channels * c.kernel_size[0] * c.kernel_size[1]
        assert abs(c.weight.std().item() / math.sqrt(2.0 / fan_out) - 1.0) < 0.05                  # kaiming_normal_(mode='fan_out', nonlinearity='relu')
    for lin in (dec.decshape, dec.deccam, dec.joint_regs[0], dec.joint_regs[10]):
        fan_in, fan_out = lin.weight.shape[1], lin.weight.shape[0]
        bound = 0.01 * math.sqrt(6.0 / (fan_in + fan_out))                                          # xavier_uniform_(gain=0.01)
        w = lin.weight.detach()
        assert w.abs().max().item() <= bound * (1 + 1e-6) and abs(w.std().item() / (bound / math.sqrt(3.0)) - 1.0) < 0.1


def test_rccl_comm_feasibility_check_is_not_collective():
    """creating the library's communicator is collective (unique-id broadcast + ncclCommInitRank); a rank on which the NCCL-API library cannot even be
    bound must find that out WITHOUT entering a collective, so that all ranks can agree on the fallback first.  RcclComm.available() touches no process group."""
    import torch.distributed as dist
    from maed_amd.ddp import RcclComm
    assert not (dist.is_available() and dist.is_initialized())
    ok, why = RcclComm.available(lib_path="/nonexistent/librccl.so")
    assert ok is False and why
