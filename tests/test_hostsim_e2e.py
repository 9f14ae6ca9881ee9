"""One whole data-parallel train step of the default configuration -- hybrid backbone on its kernels, fused parallel-mode STE blocks,
KTD head + SMPL tail, the fused LossVideo objective, gradient arena + bucketer (world 1), FusedAdam -- end to end on the host simulator
with a tiny backbone.  A smoke test of the module-level wiring for rounds without GPU time (every piece has its own parity test)."""
import os
from functools import partial

import pytest
import torch
import torch.nn as nn

from maed_amd.ddp import FusedAdam, GradBucketer, ParamArena
from maed_amd.ktd import KTD
from maed_amd.loss import Loss
from maed_amd.resnetv2 import ResNetV2
from maed_amd.trainer import TrainStep
from maed_amd.vision_transformer import VisionTransformer

from _hostsim import patched

pytestmark = pytest.mark.skipif(os.environ.get("MAED_SLOW_TESTS") != "1", reason="~2 min on the simulator: MAED_SLOW_TESTS=1")


class TinyMAED(nn.Module):
    """maed_amd.MAED (lib/models/maed.py:52-67) with a (1,1,1) backbone at 32x32 instead of the R50 at 224x224"""

    def __init__(self, dtype, channels=(128, 256, 512)):
        super().__init__()
        bb = ResNetV2(layers=(1, 1, 1), channels=channels, in_chans=3, compute_dtype=dtype)
        self.encoder = VisionTransformer(img_size=32, patch_size=16, embed_dim=128, depth=2, num_heads=2, hybrid_backbone=bb, mlp_ratio=4,
                                         qkv_bias=True, representation_size=128, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                         st_mode="parallel", num_classes=-1, compute_dtype=dtype)
        self.decoder = KTD(feat_dim=128, hidden_dim=64)

    def forward(self, x, J_regressor=None):
        n, t = x.shape[:2]
        out = self.decoder(self.encoder(x.reshape(-1, *x.shape[2:]), seqlen=t), seqlen=t, J_regressor=J_regressor)
        return {k: v.reshape(n, t, *v.shape[1:]) for k, v in out.items()}


@pytest.mark.parametrize("mode", ["bf16", "f32:bf16x3+bwd:bf16", "bf16+device_record"])
def test_train_steps_end_to_end_on_the_simulator(mode):
    """mode "f32:bf16x3+bwd:bf16" (round 5): the accurate mode's forward (fp32 operands, split-bf16 products) with the bf16 mode's backward on bf16 twins --
    backbone (bf16 autograd graph over fp32 shadows), projection (bf16 input twin, fp32 output) and STE blocks (maed_ste_block_fwd_twin) composed"""
    from maed_amd import ops
    twin = mode.startswith("f32")
    if twin:
        ops.set_float32_matmul_precision("bf16x3")
        ops.set_float32_backward_precision("bf16")
    try:
        _train_steps(torch.float32 if twin else torch.bfloat16, twin, device_record=mode.endswith("device_record"))
    finally:
        ops.set_float32_matmul_precision("exact")
        ops.set_float32_backward_precision(None)


def _train_steps(dtype, twin, device_record=False):
    """device_record (round 6): learning rate, Adam's bias corrections and the Dropout seed come from the 32-byte record (maed_adam_step_dev / maed_dropout_dev) --
    the entry points a captured step is replayed with (maed_amd/graphed.py), here in the eager loop"""
    from maed_amd import ops
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1)
    # (twin mode needs every convolution on the library's kernels: channel counts that are multiples of 64 from the first bottleneck on)
    model = TinyMAED(dtype, (256, 512, 1024) if twin else (128, 256, 512)).train()
    n_twin = ops.TWIN_FORWARDS[0]
    N, T = 2, 2
    clip = torch.randn(N, T, 3, 32, 32, generator=g)
    tgt = dict(images=clip, kp_2d=torch.cat([torch.randn(N, T, 49, 2, generator=g) * 0.3, torch.rand(N, T, 49, 1, generator=g)], -1),
               kp_3d=torch.cat([torch.randn(N, T, 49, 3, generator=g) * 0.3, torch.ones(N, T, 49, 1)], -1),
               theta=torch.cat([torch.randn(N, T, 3, generator=g) * 0.1, torch.randn(N, T, 72, generator=g) * 0.2, torch.randn(N, T, 10, generator=g)], -1),
               w_smpl=torch.ones(N, T))
    with patched():
        arena = ParamArena(model, device=torch.device("cpu"))
        bucketer = GradBucketer(arena, model, bucket_bytes=256 << 10)
        opt = FusedAdam(arena, lr=1e-3, bucketer=bucketer, model=model)
        step = TrainStep(model, Loss(e_loss_weight=300.0, e_3d_loss_weight=600.0, e_pose_loss_weight=60.0, e_shape_loss_weight=0.06,
                                     e_smpl_norm_loss=1.0, e_smpl_accl_loss=0.0, device="cpu"), opt)
        totals = []
        st = ops.DeviceTrainState(torch.device("cpu")) if device_record else None
        prev = ops.DEVICE_STATE
        try:
            if st is not None:
                opt.device_state, ops.DEVICE_STATE = st, st
            for i in range(3):
                if st is not None:
                    st.begin_step(100 + i)
                    st.upload()
                total, terms = step(target_3d=tgt)
                totals.append(float(total.detach()))
            if st is not None:
                assert st.calls == 2                                   # the decoder head's two Dropout layers took their seeds from the record
        finally:
            ops.DEVICE_STATE = prev
    assert ops.TWIN_FORWARDS[0] - n_twin == (3 * 3 if twin else 0)      # per step: the backbone + two STE blocks
    assert not ops._SHADOW, "fp32 shadows must not outlive the forward"
    assert all(torch.isfinite(torch.tensor(totals))), totals
    assert totals[-1] < totals[0], totals                              # three Adam steps on one batch must reduce its loss
    assert bool(torch.isfinite(arena.flat).all())
    dead = [n for n, p in model.named_parameters() if p.grad is None or float(p.grad.abs().max()) == 0.0]
    assert not [n for n in dead if "smpl" not in n], dead              # every trainable tensor received a gradient
