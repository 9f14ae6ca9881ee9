"""The whole train step captured as one hipGraph (maed_amd/graphed.py) against the eager step that takes the same entry points (maed_adam_step_dev /
maed_dropout_dev: learning rate, bias corrections and Dropout seed from the device record): same seeds, same initial parameters -> the same loss trajectory
(to the order of the weight gradients' fp32 atomics) and the same parameters after the last step; a changed learning rate reaches the replayed Adam kernel."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(seed=0):
    import maed_amd
    from maed_amd.ddp import FusedAdam, GradBucketer, ParamArena
    from maed_amd.loss import LossVideo
    dev = torch.device("cuda", 0)
    torch.manual_seed(seed)
    model = maed_amd.MAED(num_blocks=2, num_heads=2, embed_dim=128, hidden_dim=256, img_size=64, max_seqlen=16, compute_dtype=torch.bfloat16).to(dev)
    model.train()
    arena = ParamArena(model)
    opt = FusedAdam(arena, lr=1e-4, weight_decay=1e-5, bucketer=GradBucketer(arena, model))
    gen = torch.Generator().manual_seed(5)
    n, T = 2, 4
    clip = torch.randn(n, T, 3, 64, 64, generator=gen).to(dev)
    r = lambda *s: torch.randn(*s, generator=gen)
    tgt = {k: v.to(dev) for k, v in dict(kp_2d=torch.cat([r(n, T, 49, 2) * 0.3, torch.rand(n, T, 49, 1, generator=gen)], -1),
                                         kp_3d=torch.cat([r(n, T, 49, 3) * 0.3, torch.ones(n, T, 49, 1)], -1),
                                         theta=torch.cat([r(n, T, 3) * 0.1, r(n, T, 72) * 0.2, r(n, T, 10)], -1),
                                         w_smpl=(torch.rand(n, T, generator=gen) > 0.2).float()).items()}
    crit = LossVideo(e_loss_weight=300.0, e_3d_loss_weight=600.0, e_pose_loss_weight=60.0, e_shape_loss_weight=0.06, e_smpl_norm_loss=1.0, e_smpl_accl_loss=0.0)
    return model, arena, opt, crit, clip, tgt


def _run(eager, steps=6, lr_change_at=4):
    from maed_amd.graphed import GraphedTrainStep
    model, arena, opt, crit, clip, tgt = _setup()
    torch.manual_seed(99)
    step = GraphedTrainStep(model, crit, opt, clip, tgt, warmup=2, eager=eager)
    done = 0 if eager else 2
    losses = []
    for i in range(done, steps):
        if i == lr_change_at:
            for g in opt.param_groups:
                g["lr"] = 3e-4           # what LambdaLR does between epochs (train.py:123-127)
        losses.append(float(step().detach().float().item()))
    torch.cuda.synchronize()
    params = arena.flat.detach().clone()
    count = opt.step_count
    step.close()
    return losses, params, count


def test_graph_replay_follows_the_eager_step():
    e_losses, e_params, e_count = _run(True)
    g_losses, g_params, g_count = _run(False)
    e2_losses, e2_params, _ = _run(True)                 # what two EAGER runs differ by: the weight gradients' fp32 atomics arrive in another order, bf16 amplifies
    assert e_count == g_count == 6
    noise = max(abs(a - b) / abs(a) for a, b in zip(e_losses, e2_losses))
    for a, b in zip(e_losses[2:], g_losses):
        assert abs(a - b) <= max(4 * noise, 1e-3) * abs(a), (e_losses, e2_losses, g_losses)
    # parameters after six Adam steps (two of them at the raised learning rate).  Adam moves every parameter by ~lr per step: a replay that had kept the captured
    # learning rate for the last two steps would sit 2 x 2e-4 off ON AVERAGE; run-to-run noise flips the sign of a few tiny gradients (isolated elements, not the mean)
    d = (e_params - g_params).abs().mean().item()
    d_noise = (e_params - e2_params).abs().mean().item()
    assert d <= max(3 * d_noise, 2e-5), (d, d_noise)


def test_graph_replay_draws_fresh_dropout_masks_and_costs_the_host_little():
    import time
    from maed_amd.graphed import GraphedTrainStep
    model, arena, opt, crit, clip, tgt = _setup()
    step = GraphedTrainStep(model, crit, opt, clip, tgt, warmup=2)
    a = float(step().item())
    p0 = arena.flat.detach().clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    one = time.perf_counter() - t0                 # host time of ONE replay into an idle queue
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    host = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize()
    print(f"host time per replay: {1e3 * one:.2f} ms into an idle queue, {1e3 * host:.2f} ms back to back")
    assert one < 20e-3, one
    assert (arena.flat - p0).abs().max().item() > 0      # the replays really stepped
    # the learning rate of a replay is the one in the device record, not the captured one: at lr = 0 a replay must leave every parameter where it is
    p1 = arena.flat.detach().clone()
    for g in opt.param_groups:
        g["lr"] = 0.0
    step()
    torch.cuda.synchronize()
    assert torch.equal(arena.flat, p1)
    for g in opt.param_groups:
        g["lr"] = 1e-4
    step()
    torch.cuda.synchronize()
    assert not torch.equal(arena.flat, p1)
    assert torch.isfinite(step.loss).all() and a > 0
    step.close()


def test_device_record_entry_points_equal_the_scalar_ones_bit_for_bit():
    """maed_adam_step_dev / maed_dropout_dev against maed_adam_step / maed_dropout with the same scalars as launch arguments"""
    import numpy as np
    from maed_amd import ops, _lib as L
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    n = 1000003
    p0, gr = torch.randn(n, generator=g).to(dev), torch.randn(n, generator=g).to(dev)
    m0, v0 = torch.rand(n, generator=g).to(dev) * 0.1, torch.rand(n, generator=g).to(dev) * 0.01
    lr, b1, b2, eps, wd, step = 3e-4, 0.9, 0.999, 1e-8, 1e-5, 7
    pa, ma, va = p0.clone(), m0.clone(), v0.clone()
    ops.adam_step(pa, gr, ma, va, None, lr, b1, b2, eps, wd, step, gscale=0.5)
    st = ops.DeviceTrainState(dev)
    st.set_hyper(lr, 1.0 - b1 ** step, 1.0 - b2 ** step)
    st.begin_step(4242)
    st.upload()
    pb, mb, vb = p0.clone(), m0.clone(), v0.clone()
    ops.adam_step_dev(pb, gr, mb, vb, None, st.dev, b1, b2, eps, wd, gscale=0.5)
    torch.cuda.synchronize()
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    x = torch.randn(257, 1024, generator=g).to(dev)
    ya, yb = torch.empty_like(x), torch.empty_like(x)
    call_id = 3
    seed = (4242 + call_id * 0x9E3779B97F4A7C15) % (1 << 64)
    L.check(L.lib().maed_dropout(ops._p(x), ops._p(ya), x.numel(), 0.5, seed, ops._stream()), "dropout")
    L.check(L.lib().maed_dropout_dev(ops._p(x), ops._p(yb), x.numel(), 0.5, ops._p(st.dev), call_id, ops._stream()), "dropout_dev")
    torch.cuda.synchronize()
    assert torch.equal(ya, yb) and 0.45 < (ya != 0).float().mean().item() < 0.55
