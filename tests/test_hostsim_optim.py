"""FusedAdam (maed_adam_step through the host simulator) as a drop-in for the optimizer the reference builds
(lib/utils/utils.py:127-132: torch.optim.Adam, one group per tensor; train.py:123-127: LambdaLR; trainer.py:330-368:
checkpoints carry optimizer.state_dict()).  Also LayerNorm forward/backward kernels against ATen on the simulator."""
import copy

import torch
import torch.nn as nn

from maed_amd.ddp import FusedAdam, ParamArena

from _hostsim import patched


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.decoder = nn.Linear(7, 5)                       # arena order (forward-order key) differs from definition order
        self.encoder = nn.ModuleDict({"blocks": nn.ModuleList([nn.Linear(6, 7), nn.Linear(7, 7)])})

    def forward(self, x):
        for b in self.encoder["blocks"]:
            x = torch.tanh(b(x))
        return self.decoder(x)


def reference_optimizer(model, lr, wd):
    return torch.optim.Adam(lr=lr, params=[{"params": p, "name": n} for n, p in model.named_parameters()], weight_decay=wd)


def test_fused_adam_matches_torch_adam_with_lambda_lr_and_checkpoints():
    torch.manual_seed(0)
    ref_model = Tiny()
    model = copy.deepcopy(ref_model)
    x, y = torch.randn(16, 6), torch.randn(16, 5)
    warm = lambda epoch: (epoch + 1) * 0.2 if epoch < 3 else 0.1 ** len([m for m in (4,) if m <= epoch])   # train.py:123
    ref_opt = reference_optimizer(ref_model, 1e-2, 1e-3)
    ref_sched = torch.optim.lr_scheduler.LambdaLR(ref_opt, lr_lambda=warm)
    with patched():
        arena = ParamArena(model)
        opt = FusedAdam(arena, lr=1e-2, weight_decay=1e-3, model=model)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=warm)
        assert [g["name"] for g in opt.param_groups] == [n for n, _ in model.named_parameters()]
        for epoch in range(6):
            for m, o in ((ref_model, ref_opt), (model, opt)):
                o.zero_grad()
                ((m(x) - y) ** 2).mean().backward()
                o.step()
            ref_sched.step()
            sched.step()
            assert abs(opt.param_groups[0]["lr"] - ref_opt.param_groups[0]["lr"]) < 1e-12
            if epoch == 2:   # checkpoint hand-over in both directions, mid-run
                sd_ref, sd = ref_opt.state_dict(), opt.state_dict()
                assert sd["param_groups"][0].keys() >= {"lr", "betas", "eps", "weight_decay", "params", "name"}
                assert [g["params"] for g in sd["param_groups"]] == [g["params"] for g in sd_ref["param_groups"]]
                for i in sd_ref["state"]:
                    assert torch.allclose(sd["state"][i]["exp_avg"], sd_ref["state"][i]["exp_avg"], rtol=1e-5, atol=1e-8)
                    assert float(sd["state"][i]["step"]) == float(sd_ref["state"][i]["step"])
                ref_opt.load_state_dict(copy.deepcopy(sd))          # torch Adam resumes from OUR state
                opt.load_state_dict(copy.deepcopy(sd_ref))          # we resume from torch Adam's state
                assert opt.step_count == 3
        for (n, p), (_, q) in zip(model.named_parameters(), ref_model.named_parameters()):
            assert torch.allclose(p, q, rtol=2e-5, atol=2e-7), n


def test_fused_adam_per_group_hyperparameters_slow_path():
    torch.manual_seed(1)
    ref_model = Tiny()
    model = copy.deepcopy(ref_model)
    x, y = torch.randn(8, 6), torch.randn(8, 5)
    ref_opt = reference_optimizer(ref_model, 1e-2, 0.0)
    ref_opt.param_groups[1]["lr"] = 3e-3
    with patched():
        opt = FusedAdam(ParamArena(model), lr=1e-2, model=model)
        opt.param_groups[1]["lr"] = 3e-3
        for _ in range(3):
            for m, o in ((ref_model, ref_opt), (model, opt)):
                o.zero_grad()
                ((m(x) - y) ** 2).mean().backward()
                o.step()
    for (n, p), (_, q) in zip(model.named_parameters(), ref_model.named_parameters()):
        assert torch.allclose(p, q, rtol=2e-5, atol=2e-7), n


def test_layernorm_kernels_on_simulator():
    from maed_amd import ops
    torch.manual_seed(2)
    rows, C = 37, 128
    x = torch.randn(rows, C, requires_grad=True)
    g, b = torch.randn(C, requires_grad=True), torch.randn(C, requires_grad=True)
    dres = torch.randn(rows, C)
    y_ref = torch.nn.functional.layer_norm(x, (C,), g, b, 1e-6)
    dy = torch.randn(rows, C)
    y_ref.backward(dy)
    with patched():
        y, mean, rstd = ops.layernorm_fwd(x.detach(), g.detach(), b.detach(), torch.float32)
        dx, dg, db = ops.layernorm_bwd(dy, x.detach(), g.detach(), mean, rstd, dres=dres)
    assert torch.allclose(y, y_ref.detach(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(dx, x.grad + dres, rtol=1e-4, atol=1e-5)
    assert torch.allclose(dg, g.grad, rtol=1e-4, atol=1e-4) and torch.allclose(db, b.grad, rtol=1e-4, atol=1e-4)


def test_weight_cache_batched_refresh_on_simulator():
    """ops.WeightCache: all registered caches of a (device, dtype) are refreshed by ONE maed_weight_refresh launch into
    persistent buffers; stale detection through FusedAdam's epoch, in-place edits and re-homed storage."""
    from maed_amd import ops
    torch.manual_seed(4)
    w1, w2, w3 = torch.randn(70, 130), torch.randn(64, 64), torch.randn(5, 200)   # ragged 64x64 tiles
    with patched() as lib:
        calls = []
        real = lib.maed_weight_refresh

        def counting(*a):
            calls.append(a[1])          # entries refreshed by this launch
            return real(*a)
        lib.maed_weight_refresh = counting
        try:
            ca, cb = ops.WeightCache(), ops.WeightCache()
            (a1c, a1t), (a2c, a2t) = ca.get([w1, w2], torch.bfloat16)
            assert calls and calls[-1] >= 2
            (b3c, b3t), = cb.get([w3], torch.bfloat16)
            for w, c, t in ((w1, a1c, a1t), (w2, a2c, a2t), (w3, b3c, b3t)):
                assert torch.equal(c, w.bfloat16()) and torch.equal(t, w.bfloat16().t().contiguous())
            n = len(calls)
            ca.get([w1, w2], torch.bfloat16); cb.get([w3], torch.bfloat16)
            assert len(calls) == n, "fresh caches must not relaunch"
            ptr = a1t.data_ptr()
            w1.mul_(2.0); w3.add_(1.0)                          # optimizer-like in-place update + epoch bump
            ops.bump_weight_epoch()
            (a1c, a1t), _ = ca.get([w1, w2], torch.bfloat16)
            assert len(calls) == n + 1 and calls[-1] == 3, "one launch refreshes every registered cache"
            assert a1t.data_ptr() == ptr, "persistent buffers are reused"
            (b3c, b3t), = cb.get([w3], torch.bfloat16)
            assert len(calls) == n + 1, "the second cache was refreshed by the same launch"
            assert torch.equal(a1c, w1.bfloat16()) and torch.equal(b3t, w3.bfloat16().t().contiguous())
            # full 64 x 64 tiles of matrices with 4-element-aligned extents take the vectorised path (several tiles; full and ragged tiles in one matrix)
            cv = ops.WeightCache()
            w4, w5 = torch.randn(128, 192), torch.randn(68, 132)
            for dt in (torch.bfloat16, torch.float32):
                (v4c, v4t), (v5c, v5t) = cv.get([w4, w5], dt)
                for w, c, t in ((w4, v4c, v4t), (w5, v5c, v5t)):
                    assert torch.equal(c, w.to(dt)) and torch.equal(t, w.to(dt).t().contiguous())
            # parity mode: the [out,in] image is the fp32 master itself, only the transposed copy is built
            cf = ops.WeightCache()
            (f1, f1t), = cf.get([w2], torch.float32)
            assert f1.data_ptr() == w2.data_ptr() and torch.equal(f1t, w2.t().contiguous())
        finally:
            lib.maed_weight_refresh = real


class TinyWithUnused(Tiny):
    def __init__(self):
        super().__init__()
        self.encoder["spare"] = nn.Linear(7, 7)               # a parameter pair that never takes part in the forward (attn.ts_attn in the non-parallel st_modes)


def test_fused_adam_skips_parameters_without_a_gradient_like_torch_adam_and_orders_groups_by_definition():
    """torch.optim.Adam leaves a parameter whose .grad is None alone -- no weight decay, no moment update.  The arena's gradients are never None, so
    FusedAdam takes "received no gradient this step" from the bucketer's readiness reports and steps the runs of active tensors only.  Also: without model=
    the parameter groups follow model.named_parameters() order (the reference optimizer's), not the arena's forward order."""
    from maed_amd.ddp import GradBucketer
    torch.manual_seed(1)
    ref_model = TinyWithUnused()
    model = copy.deepcopy(ref_model)
    x, y = torch.randn(8, 6), torch.randn(8, 5)
    ref_opt = reference_optimizer(ref_model, 1e-2, 0.1)
    with patched():
        arena = ParamArena(model)
        opt = FusedAdam(arena, lr=1e-2, weight_decay=0.1, bucketer=GradBucketer(arena, model))
        assert [g["name"] for g in opt.param_groups] == [n for n, _ in model.named_parameters()]
        for _ in range(3):
            for m, o in ((ref_model, ref_opt), (model, opt)):
                o.zero_grad(set_to_none=True) if o is ref_opt else o.zero_grad()
                ((m(x) - y) ** 2).mean().backward()
                o.step()
    for (n, p), (_, q) in zip(model.named_parameters(), ref_model.named_parameters()):
        assert torch.allclose(p, q, rtol=2e-5, atol=2e-7), n
    torch.manual_seed(1)
    fresh = TinyWithUnused()
    for (n, p), (_, q) in zip(model.named_parameters(), fresh.named_parameters()):
        if "spare" in n:
            assert torch.equal(p, q), f"{n} must not have been touched (weight decay 0.1 would have shrunk it)"


def test_fused_adam_with_device_record_equals_host_scalars_and_follows_the_schedule():
    """maed_adam_step_dev (round 6: learning rate and bias corrections read from the 32-byte device record, include/maed_hip.h maed_train_state -- what lets a captured
    step be replayed) against maed_adam_step with the same scalars as launch arguments: bit-identical parameters over six steps under a LambdaLR schedule"""
    from maed_amd import ops
    torch.manual_seed(1)
    m_host = Tiny()
    m_dev = copy.deepcopy(m_host)
    x, y = torch.randn(16, 6), torch.randn(16, 5)
    warm = lambda epoch: (epoch + 1) * 0.25 if epoch < 3 else 0.5
    with patched():
        opts = []
        for m, with_state in ((m_host, False), (m_dev, True)):
            arena = ParamArena(m)
            opt = FusedAdam(arena, lr=1e-2, weight_decay=1e-3, model=m)
            if with_state:
                opt.device_state = ops.DeviceTrainState(torch.device("cpu"))
            opts.append((m, opt, torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=warm)))
        for epoch in range(6):
            for m, opt, sched in opts:
                opt.zero_grad()
                ((m(x) - y) ** 2).mean().backward()
                opt.step()
                sched.step()
        for p, q in zip(m_host.parameters(), m_dev.parameters()):
            assert torch.equal(p, q)
        # the record holds what the last step used
        st = opts[1][1].device_state
        import numpy as np
        lr, bc1, bc2 = st.dev.numpy()[:12].view(np.float32)
        assert abs(lr - 1e-2 * warm(5)) < 1e-9 and abs(bc1 - (1 - 0.9 ** 6)) < 1e-6 and abs(bc2 - (1 - 0.999 ** 6)) < 1e-6


def test_fused_adam_with_device_record_refuses_per_group_schedules():
    import pytest
    from maed_amd import ops
    m = Tiny()
    with patched():
        opt = FusedAdam(ParamArena(m), lr=1e-2, model=m)
        opt.device_state = ops.DeviceTrainState(torch.device("cpu"))
        opt.param_groups[1]["lr"] = 5e-3
        opt.zero_grad()
        m(torch.randn(4, 6)).sum().backward()
        with pytest.raises(RuntimeError):
            opt.step()
