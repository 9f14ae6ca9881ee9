"""The backbone's GPU code path (ResNetV2.forward_features on a library device: batched weight standardisation, 1x1 convolutions on
the GEMM kernels with residual-fork fusion, fused GroupNorm(+residual+ReLU) with scratch arenas and bit masks, the max-pool kernels,
and -- with MAED_CONV3X3=own -- the implicit-GEMM 3x3 convolutions with their transposed images / fp32 dW slices) run end to end on the
host simulator in bf16, against the pure-ATen fp32 path of the same module (which tests/test_host_logic.py ties to the reference's own
ResNetV2 outputs).  This is the module-level glue that the kernel-level tests cannot see."""
import copy
import os

import pytest
import torch

from maed_amd import resnetv2
from maed_amd.resnetv2 import ResNetV2

from _hostsim import patched


def cos(a, b):
    a, b = a.detach().double().flatten(), b.detach().double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


@pytest.mark.parametrize("own3x3", [pytest.param(False, marks=pytest.mark.skipif(os.environ.get("MAED_SLOW_TESTS") != "1",
                                                                                reason="30 s; the default path's backward is in the -m gpu suite: MAED_SLOW_TESTS=1")), True])
def test_backbone_gpu_path_on_simulator_matches_aten(own3x3, monkeypatch):
    monkeypatch.setattr(resnetv2, "_OWN_CONV3X3", own3x3)
    torch.manual_seed(0)
    ref = ResNetV2(layers=(2,), channels=(256,), in_chans=3, compute_dtype=torch.float32)            # built AFTER the patch: _own3x3 follows the switch
    for m in ref._norms:                                                                               # non-trivial affine parameters
        torch.nn.init.normal_(m.weight, 1.0, 0.2); torch.nn.init.normal_(m.bias, 0.0, 0.2)
    sim = copy.deepcopy(ref)
    sim.compute_dtype = torch.bfloat16
    assert bool(sim._own3x3) == own3x3
    from maed_amd import ops
    calls = {"conv3x3": 0, "wgrad": 0}
    real_conv, real_wgrad = ops.conv3x3, ops.conv3x3_wgrad
    monkeypatch.setattr(ops, "conv3x3", lambda *a, **k: (calls.__setitem__("conv3x3", calls["conv3x3"] + 1), real_conv(*a, **k))[1])
    monkeypatch.setattr(ops, "conv3x3_wgrad", lambda *a, **k: (calls.__setitem__("wgrad", calls["wgrad"] + 1), real_wgrad(*a, **k))[1])
    x = torch.randn(4, 3, 16, 16)                  # stage 1 sees 4 x 4 x 4 = 64 pixels: one row tile, and a multiple of 64 for the own 3x3 weight gradient
    gout = torch.randn(4, 256, 4, 4)
    yr = ref(x)
    (yr * gout).sum().backward()
    with patched():
        ys = sim(x)
        assert ys.dtype == torch.bfloat16 and ys.shape == yr.shape
        (ys.float() * gout).sum().backward()
    assert cos(ys.float(), yr.detach()) > 0.999, cos(ys.float(), yr.detach())
    worst = 1.0
    for (n, p), q in zip(sim.named_parameters(), ref.parameters()):
        assert p.grad is not None, n
        c = cos(p.grad, q.grad)
        worst = min(worst, c)
        # bf16 activations and gradients through ~10 layers: the shallowest layers are the noisiest (on hardware the bf16 backbone
        # sits at cosine 0.94 against the fp64 oracle for BOTH the GEMM-convolution and the all-MIOpen path)
        assert c > 0.93, (n, c)
    print(f"own3x3={own3x3}: worst parameter-gradient cosine {worst:.4f}, library 3x3 calls {calls}")
    # two stride-1 3x3 convolutions: forward + input gradient each on maed_conv3x3_fwd, weight gradients on maed_conv3x3_wgrad
    assert calls == ({"conv3x3": 4, "wgrad": 2} if own3x3 else {"conv3x3": 0, "wgrad": 0})


class _BackboneToy(torch.nn.Module):
    """tiny hybrid backbone on its library path + an autograd-managed head: the two kinds of parameters the bucketer tracks in the real
    model's patch_embed (kernel-written conv / GroupNorm gradients reported through ResNetV2.grads_ready, autograd hooks for the rest)"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.backbone = ResNetV2(layers=(1,), channels=(256,), in_chans=3, compute_dtype=torch.bfloat16)
        self.head = torch.nn.Linear(256, 8)

    def forward(self, x):
        return self.head(self.backbone(x).float().mean(dim=(2, 3)))


def _backbone_ddp_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    from maed_amd.ddp import GradBucketer, ParamArena
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with patched():
            model = _BackboneToy()
            arena = ParamArena(model, device=torch.device("cpu"))
            bucketer = GradBucketer(arena, model, bucket_bytes=64 << 10)
            bucketer.broadcast_parameters(0)
            assert len(bucketer.buckets) > 3 and len(bucketer._fused) == len(model.backbone.fused_parameters())
            x = torch.randn(4, 3, 8, 8, generator=torch.Generator().manual_seed(5))
            shard = slice(rank * 2, rank * 2 + 2)
            for _ in range(1):                                   # (counter reset across steps: tests/test_host_logic.py, KTD variant)
                arena.zero_grad()
                (model(x[shard]) ** 2).sum().backward()
                assert all(bucketer._launched), "every bucket must have been reduced during backward"
                bucketer.finish()
            out[rank] = arena.grad.clone()
    finally:
        dist.destroy_process_group()


def test_backbone_fused_gradients_through_the_bucketer_gloo_world2():
    """2 gloo ranks x 2 images on the simulator == one process x 4 images (GroupNorm is per sample): the kernel-written backbone
    gradients land in the arena, every bucket completes during backward via ResNetV2.grads_ready, and the all-reduced sum matches"""
    import torch.multiprocessing as mp
    from maed_amd.ddp import ParamArena
    world, port = 2, 31533 + os.getpid() % 1000
    out = mp.Manager().dict()
    mp.spawn(_backbone_ddp_worker, args=(world, port, out), nprocs=world, join=True)
    assert torch.equal(out[0], out[1])
    with patched():
        ref = _BackboneToy()
        arena = ParamArena(ref, device=torch.device("cpu"))
        x = torch.randn(4, 3, 8, 8, generator=torch.Generator().manual_seed(5))
        (ref(x) ** 2).sum().backward()
    for n, o in zip(arena.names, arena.offsets):
        p = dict(ref.named_parameters())[n]
        a, b = out[0][o:o + p.numel()], arena.grad[o:o + p.numel()]
        assert cos(a, b) > 0.999, (n, cos(a, b))                 # same bf16 kernels; only the summation order differs


def test_per_stage_weight_standardisation_reports_readiness_stage_by_stage(monkeypatch):
    """MAED_WS_PER_STAGE: the backbone's gradients are reported to the bucketer in two steps -- the last stage first, as soon as its
    backward is done -- instead of once after the very last backward kernel; numbers as on the ATen path (and, with MAED_SLOW_TESTS=1,
    equal to the single batched launch)"""
    torch.manual_seed(0)
    base = ResNetV2(layers=(1, 1), channels=(256, 256), in_chans=3, compute_dtype=torch.bfloat16)
    x = torch.randn(2, 3, 16, 16)
    gout = torch.randn(2, 256, 2, 2)
    ref = copy.deepcopy(base)
    ref.compute_dtype = torch.float32
    (ref(x) * gout).sum().backward()                                         # ATen fp32 path (CPU tensors outside patched())

    def run(per_stage):
        monkeypatch.setattr(resnetv2, "_WS_PER_STAGE", per_stage)
        m = copy.deepcopy(base)
        reports = []
        m.grads_ready = lambda owner: reports.append(owner)
        with patched():
            (m(x).float() * gout).sum().backward()
        return [p.grad.clone() for p in m.parameters()], reports, m

    g1, rep1, m1 = run(True)
    assert rep1 == [m1._ws_groups[1], m1._ws_groups[0]]                      # last stage first, then stem + stage 0
    fused = [id(p) for grp in rep1 for p in grp.fused_parameters()]
    assert sorted(fused) == sorted(id(p) for p in m1.fused_parameters()) and len(set(fused)) == len(fused)      # every parameter exactly once
    for a, q, (n, _) in zip(g1, ref.parameters(), base.named_parameters()):
        assert cos(a, q.grad) > 0.93, (n, cos(a, q.grad))
    if os.environ.get("MAED_SLOW_TESTS") == "1":
        g0, rep0, m0 = run(False)
        assert rep0 == [m0]                                                  # one report, after the batched backward
        for a, b, (n, _) in zip(g0, g1, base.named_parameters()):
            assert cos(a, b) > 0.9999, (n, cos(a, b))


def test_switching_gemm_convs_off_at_run_time_also_drops_their_direct_hand_over():
    """tests/test_gpu_model.py's all-MIOpen variant sets `net._gemm_convs = []` on a live model: WeightStdFn must then stop producing
    transposed images / fp32 slices for them (it reads `_direct_convs`, which therefore has to follow `_gemm_convs`)"""
    net = ResNetV2(layers=(1, 2, 1), channels=(256, 512, 1024), compute_dtype=torch.bfloat16)
    assert len(net._gemm_convs) == 11 and net._direct_convs == net._gemm_convs + net._own3x3      # 9 stride-1 + the 2 stride-2 downsample shortcuts
    net._gemm_convs = []
    assert net._direct_convs == net._own3x3
    assert all(g._direct_convs == [] for g in net._ws_groups) or bool(net._own3x3)


def test_convolution_epilogues_feed_the_groupnorm_statistics(monkeypatch):
    """>= 128 pixels per frame: every library convolution (1x1 GEMM, own 3x3) hands its GroupNorm the statistics; forward and gradients equal
    the separate-statistics-pass variant (MAED_GN_FUSE_STATS=0) up to fp32 summation order"""
    from maed_amd import ops
    torch.manual_seed(0)
    net = ResNetV2(layers=(1,), channels=(256,), in_chans=3, compute_dtype=torch.bfloat16)
    for m in net._norms:
        torch.nn.init.normal_(m.weight, 1.0, 0.2); torch.nn.init.normal_(m.bias, 0.0, 0.2)
    x = torch.randn(2, 3, 48, 48)                     # stem /4 -> 12 x 12 = 144 pixels per frame: a 128-row tile straddles the two frames
    gout = torch.randn(2, 256, 12, 12)
    seen = []
    real = ops.GroupNormFn.forward
    monkeypatch.setattr(ops.GroupNormFn, "forward", staticmethod(lambda ctx, *a: (seen.append(bool(a[9]) if len(a) > 9 else False), real(ctx, *a))[1]))
    res = {}
    for fuse in (True, False):
        monkeypatch.setattr(resnetv2, "_FUSE_GN_STATS", fuse)
        seen.clear()
        net.zero_grad()
        for p in net.parameters():
            p.grad = None
        with patched():
            y = net(x)
            (y.float() * gout).sum().backward()
        res[fuse] = (y.float().clone(), [p.grad.clone() for p in net.parameters()], list(seen))
    # stem norm follows the 7x7 library (MIOpen/ATen) convolution: separate pass; the block's three convolutions + the downsample one are fused
    assert res[True][2].count(True) == len(net._norms) - 1 and not any(res[False][2]), res[True][2]
    assert torch.allclose(res[True][0], res[False][0], rtol=2e-2, atol=2e-2) and cos(res[True][0], res[False][0]) > 0.9999
    for a, b in zip(res[True][1], res[False][1]):
        assert cos(a, b) > 0.999


def test_two_stage_backbone_downsample_shortcut_runs_on_packed_pixels():
    """stage 2's 1x1 stride-2 downsample convolution is a library GEMM convolution (ops.Conv1x1Fn stride=2): forward and every parameter
    gradient of a two-stage backbone against the fp32 ATen composition"""
    from maed_amd import ops
    torch.manual_seed(0)
    ref = ResNetV2(layers=(1, 1), channels=(256, 512), in_chans=3, compute_dtype=torch.float32)
    for m in ref._norms:
        torch.nn.init.normal_(m.weight, 1.0, 0.2); torch.nn.init.normal_(m.bias, 0.0, 0.2)
    sim = copy.deepcopy(ref)
    sim.compute_dtype = torch.bfloat16
    strided = [i for i in sim._gemm_convs if sim._convs[i].stride == (2, 2)]
    assert len(strided) == 1 and strided[0] in sim._direct_convs
    x = torch.randn(1, 3, 32, 32)
    yr = ref(x)
    gout = torch.randn_like(yr)
    (yr * gout).sum().backward()
    seen = []
    real = ops.Conv1x1Fn.forward
    with patched():
        ops.Conv1x1Fn.forward = staticmethod(lambda ctx, *a: (seen.append(a[6] if len(a) > 6 else 1), real(ctx, *a))[1])
        try:
            ys = sim(x)
            (ys.float() * gout).sum().backward()
        finally:
            ops.Conv1x1Fn.forward = real
    assert seen.count(2) == 1, seen
    assert cos(ys.float(), yr.detach()) > 0.999
    for (n, p), q in zip(sim.named_parameters(), ref.parameters()):
        assert p.grad is not None and cos(p.grad, q.grad) > 0.93, (n, cos(p.grad, q.grad))


# (the process-wide "bf16x6" runs the same kernels as "module:bf16x6" forward and as "bf16x3" backward; kernel-level bf16x6 cases: test_hostsim_x3.py)
@pytest.mark.parametrize("mode,tol_y,tol_g", [("bf16x3", 2e-4, 2e-3), ("module:bf16x6", 2e-5, 2e-4)])
def test_backbone_f32_split_mode_runs_on_library_convolutions(mode, tol_y, tol_g, monkeypatch):
    """compute_dtype=float32 with the split-bf16 matmul mode: the fp32 backbone takes the bf16 mode's code path -- 1x1 convolutions on
    maed_gemm_nt / maed_conv1x1_fwd (+ GroupNorm statistics from the epilogue), 3x3 on the implicit-GEMM kernel incl. its transposed-image input
    gradient, weight gradients on the TN kernels, stride-2 shortcut on packed pixels -- and agrees with the fp32 ATen composition to the
    scheme's error level (relative to the tensor maximum), orders of magnitude tighter than the bf16 mode's cosine 0.93"""
    from maed_amd import ops
    torch.manual_seed(0)
    ref = ResNetV2(layers=(1, 1), channels=(256, 512), in_chans=3, compute_dtype=torch.float32)
    for m in ref._norms:
        torch.nn.init.normal_(m.weight, 1.0, 0.2); torch.nn.init.normal_(m.bias, 0.0, 0.2)
    sim = copy.deepcopy(ref)
    if mode.startswith("module:"):                    # the process-wide mode stays exact; the backbone carries its own engine (MAED_F32X6 dtype codes per call)
        sim.f32_matmul, mode = mode.split(":")[1], "exact"
    x = torch.randn(1, 3, 64, 64)                     # stem /4 -> 16 x 16 = 256 pixels per frame (GroupNorm statistics from the epilogues)
    yr = ref(x)
    gout = torch.randn_like(yr)
    (yr * gout).sum().backward()
    calls = {"c1": 0, "c3": 0, "wg3": 0, "tn": 0}
    real = dict(c1=ops.Conv1x1Fn.forward, c3=ops.conv3x3, wg3=ops.conv3x3_wgrad, tn=ops.gemm_tn_wgrad)
    monkeypatch.setattr(ops, "conv3x3", lambda *a, **k: (calls.__setitem__("c3", calls["c3"] + 1), real["c3"](*a, **k))[1])
    monkeypatch.setattr(ops, "conv3x3_wgrad", lambda *a, **k: (calls.__setitem__("wg3", calls["wg3"] + 1), real["wg3"](*a, **k))[1])
    monkeypatch.setattr(ops, "gemm_tn_wgrad", lambda *a, **k: (calls.__setitem__("tn", calls["tn"] + 1), real["tn"](*a, **k))[1])
    old = ops.get_float32_matmul_precision()
    try:
        ops.set_float32_matmul_precision(mode)
        with patched():
            ops.Conv1x1Fn.forward = staticmethod(lambda ctx, *a: (calls.__setitem__("c1", calls["c1"] + 1), real["c1"](ctx, *a))[1])
            try:
                ys = sim(x)
                assert ys.dtype == torch.float32
                (ys * gout).sum().backward()
            finally:
                ops.Conv1x1Fn.forward = real["c1"]
    finally:
        ops.set_float32_matmul_precision(old)
    # 2 blocks x (conv1, conv3, downsample) on the GEMM path; the stride-1 3x3 of stage 1: forward + input gradient + weight gradient on the library
    # (stage 2's stride-2 3x3: library forward, framework backward)
    assert calls["c1"] == 6 and calls["c3"] == 3 and calls["wg3"] == 1 and calls["tn"] == 6, calls
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
    assert rel(ys, yr.detach()) <= tol_y, rel(ys, yr.detach())
    worst = max(rel(p.grad, q.grad) for p, q in zip(sim.parameters(), ref.parameters()))
    print(f"{mode}: output rel-to-max {rel(ys, yr.detach()):.2e}, worst parameter gradient rel-to-max {worst:.2e}")
    for (n, p), q in zip(sim.named_parameters(), ref.parameters()):
        assert p.grad is not None and rel(p.grad, q.grad) <= tol_g, (n, rel(p.grad, q.grad))


@pytest.mark.parametrize("own", [True, False])
def test_stem_convolution_on_the_library_when_the_frame_geometry_allows(own, monkeypatch):
    """round 4: 32 x 32 frames (16 x 16 = 256 output pixels, 16 per row) take ops.StemConvFn -- padded 4-slot image, statistics of the stem norm from the
    epilogue, weight gradient into the fp32 slice WeightStdFn hands out; MAED_STEM_OWN=0 (and geometries the kernels do not cover: every other test of this
    file with 16 x 16 / 48 x 48 frames) stay on the framework convolution.  Forward and every parameter gradient against the fp32 ATen composition."""
    from maed_amd import ops
    monkeypatch.setenv("MAED_STEM_OWN", "1" if own else "0")
    torch.manual_seed(3)
    ref = ResNetV2(layers=(1,), channels=(256,), in_chans=3, compute_dtype=torch.float32)
    for m in ref._norms:
        torch.nn.init.normal_(m.weight, 1.0, 0.2); torch.nn.init.normal_(m.bias, 0.0, 0.2)
    sim = copy.deepcopy(ref)
    sim.compute_dtype = torch.bfloat16
    x = torch.randn(2, 3, 32, 32)
    yr = ref(x)
    gout = torch.randn_like(yr)
    (yr * gout).sum().backward()
    calls = []
    real = ops.StemConvFn.forward
    with patched():
        ops.StemConvFn.forward = staticmethod(lambda ctx, *a: (calls.append((tuple(a[0].shape), a[2] is not None, a[3] is not None)), real(ctx, *a))[1])
        try:
            ys = sim(x)
            (ys.float() * gout).sum().backward()
        finally:
            ops.StemConvFn.forward = real
    assert calls == ([((2, 4, 37, 38), True, True)] if own else []), calls
    assert sim._own_stem_now == [] and sim.stem.conv._prepadded is False
    assert cos(ys.float(), yr.detach()) > 0.999
    for (n, p), q in zip(sim.named_parameters(), ref.parameters()):
        assert p.grad is not None and cos(p.grad, q.grad) > 0.97, (n, cos(p.grad, q.grad))


@pytest.mark.parametrize("per_stage", [False, True])
def test_backbone_twin_mode_fp32_forward_on_shadows_bf16_backward_on_twins(per_stage, monkeypatch):
    """round 5, ops.set_float32_backward_precision("bf16"): a compute_dtype = float32 backbone in a training pass runs the bf16 mode's AUTOGRAD GRAPH (own stem,
    GEMM / implicit-GEMM convolutions, fused GroupNorm, bit masks, fp32 dW slices) over bf16 twins while the forward chain computes on fp32 shadows with the
    split-bf16 products: the output's shadow agrees with the fp32 ATen composition at the split engine's level (the bf16 mode: cosine 0.999), the twin is its bf16
    rounding, every parameter gradient is what the bf16 mode delivers, and a no-grad pass of the same module stays the plain fp32 path."""
    from maed_amd import ops
    # per_stage: the data-parallel path's weight standardisation per backbone stage (resnetv2._WS_PER_STAGE) -- one bf16 autograd node + one fp32 shadow image set per stage
    monkeypatch.setattr(resnetv2, "_WS_PER_STAGE", per_stage)
    torch.manual_seed(5)
    ref = ResNetV2(layers=(1, 1), channels=(256, 512), in_chans=3, compute_dtype=torch.float32)
    for m in ref._norms:
        torch.nn.init.normal_(m.weight, 1.0, 0.2); torch.nn.init.normal_(m.bias, 0.0, 0.2)
    sim = copy.deepcopy(ref)
    x = torch.randn(2, 3, 64, 64)
    yr = ref(x)
    gout = torch.randn_like(yr)
    (yr * gout).sum().backward()
    old = ops.get_float32_matmul_precision()
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
    try:
        ops.set_float32_matmul_precision("bf16x3")
        ops.set_float32_backward_precision("bf16")
        with patched():
            n0 = ops.TWIN_FORWARDS[0]
            ys = sim(x)
            assert ops.TWIN_FORWARDS[0] == n0 + 1 and ys.dtype == torch.bfloat16
            y32 = ops.shadow_of(ys)
            assert y32 is not None and y32.dtype == torch.float32
            assert rel(y32, yr.detach()) <= 2e-4, rel(y32, yr.detach())
            assert torch.equal(ys, y32.to(torch.bfloat16))
            ops.shadow_clear()
            (ys.float() * gout).sum().backward()
            assert sim._own_stem_now == [] and sim.stem.conv._prepadded is False
            with torch.no_grad():
                yn = sim(x)
            assert yn.dtype == torch.float32 and ops.TWIN_FORWARDS[0] == n0 + 1 and rel(yn, yr.detach()) <= 2e-4
    finally:
        ops.set_float32_matmul_precision(old)
        ops.set_float32_backward_precision(None)
    worst = 1.0
    for (n, p), q in zip(sim.named_parameters(), ref.parameters()):
        assert p.grad is not None, n
        c = cos(p.grad, q.grad)
        worst = min(worst, c)
        assert c > 0.97, (n, c)
    print(f"twin mode: output shadow rel-to-max {rel(y32, yr.detach()):.2e}, worst parameter-gradient cosine {worst:.4f}")


def test_a_failing_forward_leaves_no_prepadded_stem_behind(monkeypatch):
    """forward_features marked the stem "pre-padded" before entering its try/finally; an exception in between (alignment, out of memory, an unsupported size)
    left the mark set, and the next forward on another route convolved an UNPADDED image -- silently wrong.  Every hand-over slot is now set inside the try."""
    from maed_amd import ops
    torch.manual_seed(2)
    ref = ResNetV2(layers=(1,), channels=(256,), in_chans=3, compute_dtype=torch.float32)
    sim = copy.deepcopy(ref)
    sim.compute_dtype = torch.bfloat16
    x = torch.randn(2, 3, 32, 32)
    real = ops.WeightStdFn.apply

    def boom(*a, **k):
        raise RuntimeError("simulated failure after the stem input was prepared")
    with patched():
        monkeypatch.setattr(ops.WeightStdFn, "apply", boom)
        with pytest.raises(RuntimeError, match="simulated failure"):
            sim(x)
        monkeypatch.setattr(ops.WeightStdFn, "apply", real)
    assert sim.stem.conv._prepadded is False and sim.stem.conv._stem_hw is None and sim._own_stem_now == []
    sim.compute_dtype = torch.float32
    with torch.no_grad():                      # the ATen route of the same module: a stale mark would skip the TF-SAME padding here
        assert torch.allclose(sim(x), ref(x), rtol=1e-5, atol=1e-5)
