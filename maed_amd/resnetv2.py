"""Hybrid R50 backbone of the STE (reference: lib/models/resnetv2.py).

Not torchvision's ResNet-50: the ViT-hybrid ResNetV2 with 3 stages (3,4,9), weight-standardised
convolutions with TF 'SAME' padding, GroupNorm(32) and non-pre-activation bottlenecks, output
stride 16 (resnetv2.py:74-93,159-204,277-335; vision_transformer.py:564-566).

Per BASELINE.json's north_star the convolutions ride on MIOpen through PyTorch-ROCm; what this
module decides is the MI355X-side data layout: activations are channels_last in the compute dtype
(bf16 for throughput), the weight standardisation is done once per forward in fp32 (the reference
recomputes it twice per conv call, resnetv2.py:92-93) and the output is handed to the 1x1
projection GEMM as a (F*H*W, C) row-major matrix without a transpose copy.

Module / parameter names match the reference so its checkpoints load unchanged.
"""
import math
import os
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


# The 3x3 convolutions run on the library's implicit-GEMM kernel (ops.Conv3x3Fn): measured against MIOpen on MI355X at the cfg3 layer
# shapes (profiles/r02_call1_conv3x3_micro.txt) forward 0.79 vs 1.71 ms, input gradient 0.66 vs 1.29 ms, weight gradient 1.09 vs
# 1.44 ms per step; the train step 28.14 -> 26.38 ms.  MAED_CONV3X3=miopen switches back (A/B knob).
_OWN_CONV3X3 = os.environ.get("MAED_CONV3X3", "own") == "own"


# MAED_WS_PER_STAGE=1: weight standardisation launched per backbone stage, right before the stage runs, instead of once for all 53
# convolutions.  Single-GPU cost: two more (tiny) launches per direction.  Purpose: data-parallel overlap -- a stage's convolution
# weight gradients become final, and its gradient bucket starts its all-reduce, as soon as THAT stage's backward is done; with one
# batched launch the whole backbone (47 MB at cfg3) is reported only by the very last kernel of the backward and reduced un-overlapped.
# Measured on one MI355X (profiles/r02_call2_steady_*.csv): +0.13 ms per step (ws_fwd 0.143 -> 0.160, ws_bwd 0.099 -> 0.145 ms, two more
# fills).  Default "auto": per stage when the process is one rank of a data-parallel job (torch.distributed initialised with world > 1),
# one batched launch otherwise; MAED_WS_PER_STAGE=0 / 1 forces either.
_WS_PER_STAGE_ENV = os.environ.get("MAED_WS_PER_STAGE", "auto")

# identity blocks: masked residual gradient applied by the consumer instead of written by the GroupNorm backward (MAED_GN_LAZY_DRES=0: materialise it; A/B knob)
_LAZY_RES_GRAD = os.environ.get("MAED_GN_LAZY_DRES", "1") == "1"

# GroupNorm statistics accumulated in the epilogue of the convolution in front (maed_conv1x1_fwd / maed_conv3x3_fwd gn_sums) instead of
# a separate pass over the activation (52 launches, 0.64 ms per step at cfg3).  MAED_GN_FUSE_STATS=0 switches back (A/B knob).
_FUSE_GN_STATS = os.environ.get("MAED_GN_FUSE_STATS", "1") == "1"


def _ws_per_stage():
    if _WS_PER_STAGE is not None:
        return _WS_PER_STAGE
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


_WS_PER_STAGE = None if _WS_PER_STAGE_ENV == "auto" else _WS_PER_STAGE_ENV == "1"     # (tests monkeypatch this to True / False)


class _WsGroup:
    """the convolutions / norms of one backbone stage as the `owner` ops.WeightStdFn talks to (same protocol as ResNetV2 itself)"""

    def __init__(self, parent, conv_idx, norms):
        self.parent, self.conv_idx, self.norms = parent, list(conv_idx), list(norms)
        self._pending_backwards = 0
        self._ws_on_side = False
        self._w_std_t, self._dw_slices, self._dw_arena = {}, {}, None

    @property
    def f32_matmul(self):
        return self.parent.f32_matmul

    def conv_weights(self):
        return [self.parent._convs[i].weight for i in self.conv_idx]

    def fused_parameters(self):
        return self.conv_weights() + [t for m in self.norms for t in (m.weight, m.bias)]

    @property
    def _direct_convs(self):
        pos = {ci: k for k, ci in enumerate(self.conv_idx)}
        return [pos[i] for i in self.parent._direct_convs if i in pos]

    @property
    def grads_ready(self):
        cb = self.parent.grads_ready
        return None if cb is None else (lambda _owner: cb(self))          # the bucketer marks exactly this stage's parameters


def _same_pad(x, k, s, value=0.0):
    """TF 'SAME' padding computed from the input size (resnetv2.py:51-59): left = pad//2."""
    ih, iw = x.shape[-2:]
    ph = max((math.ceil(ih / s) - 1) * s + k - ih, 0)
    pw = max((math.ceil(iw / s) - 1) * s + k - iw, 0)
    if ph > 0 or pw > 0:
        x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2], value=value)
    return x


class StdConv2dSame(nn.Conv2d):
    """resnetv2.py:74-93: biased std, eps added to the std, no bias."""

    def __init__(self, in_channel, out_channels, kernel_size, stride=1, dilation=1, groups=1, bias=False, eps=1e-5):
        super().__init__(in_channel, out_channels, kernel_size, stride=stride, padding=0, dilation=dilation, groups=groups, bias=bias)
        self.eps = eps

    def get_weight(self):
        std, mean = torch.std_mean(self.weight, dim=[1, 2, 3], keepdim=True, unbiased=False)
        return (self.weight - mean) / (std + self.eps)

    _w_std = None  # set by ResNetV2 for the duration of a forward: weight standardised by the batched HIP kernel
    _w_t = None    # 1x1 stride-1 convolutions in bf16 mode: transposed standardised weight (I, O) -> the GEMM path
    _dw = None     # ... and the fp32 slice their weight gradient accumulates into
    _prec = None   # fp32 matrix-product engine of the owning backbone (ResNetV2.f32_matmul), None = the process-wide mode
    _prepadded = False   # set by ResNetV2 for the stem when ops.stem_input already applied the TF-SAME padding ("own": ... in the layout of maed_stem7x7s2_*)
    _gn_behind = None    # stem only: (GroupNormAct,) it feeds -- a tuple, so that the norm is not registered a second time as a submodule
    _stem_hw = None      # stem, "own" route: (H, W) of the unpadded frames

    @staticmethod
    def _gn_sums_for(gn, x_shape, out_channels, stride):
        """the pre-zeroed statistics slice of the GroupNorm that follows this convolution, if the convolution's epilogue may fill it
        (library kernels only: 32 groups of 2^k >= 2 channels, >= 128 output pixels per frame); marks the norm so that it skips its own
        statistics pass"""
        if gn is None or not _FUSE_GN_STATS or gn._sums_buf is None or gn.num_groups != 32:
            return None
        cpg = out_channels // 32
        hw = (-(-x_shape[-2] // stride)) * (-(-x_shape[-1] // stride))
        if out_channels % 32 or cpg < 2 or cpg & (cpg - 1) or hw < 128 or x_shape[0] * x_shape[1] * x_shape[2] * x_shape[3] * 2 >= 1 << 32:     # (bf16 kernels: 32-bit byte offsets)
            return None
        gn._stats_ready = True
        return gn._sums_buf

    def forward(self, x, fork=False, gn=None, lazy_short=False):
        """fork=True (GEMM convolutions only): returns (conv(x), alias of x) -- see ops.Conv1x1Fn.
        gn: the GroupNormAct this convolution feeds -- its statistics are then accumulated by the convolution's epilogue"""
        w = self._w_std
        if w is not None and self._w_t is not None and self.kernel_size == (1, 1):
            # 1x1, stride 1, bf16: three GEMMs on libmaed_hip instead of MIOpen's implicit-GEMM solvers (which zero-fill the
            # output and cast weight gradients through an fp32 workspace first): ops.Conv1x1Fn
            return ops.Conv1x1Fn.apply(x, w, self._w_t, self._dw, fork, self._gn_sums_for(gn, x.shape, self.out_channels, self.stride[0]), self.stride[0], lazy_short,
                                       self._prec)
        assert not fork
        if (w is not None and _OWN_CONV3X3 and self.kernel_size == (3, 3) and ops.on_library_device(x) and ops.lib_matmul_dtype(x.dtype, self._prec)
                and self.in_channels % 64 == 0 and self.out_channels % 8 == 0 and self.dilation == (1, 1) and self.groups == 1):
            # implicit-GEMM forward / stride-1 input gradient on libmaed_hip instead of MIOpen
            # (_w_t / _dw: transposed image and fp32 dW slice from WeightStdFn for the stride-1 ones, see ResNetV2._own3x3)
            return ops.Conv3x3Fn.apply(x, w, self.stride[0], self._w_t, self._dw, self._gn_sums_for(gn, x.shape, self.out_channels, self.stride[0]), self._prec)
        if w is None:  # stand-alone use / CPU: per-conv ATen composition
            w = self.get_weight().to(x.dtype)
            if ops.on_library_device(x):
                w = w.contiguous(memory_format=torch.channels_last)
        if self._prepadded == "own":
            # the stem on the library (ops.StemConvFn): x is the 4-channel padded image of ops.stem_input(own=True); statistics of the norm behind from the epilogue
            gn = self._gn_behind[0] if self._gn_behind else None
            sums = None
            # (a shadowed input -- the "bf16" backward mode's fp32 forward -- runs the vendor's fp32 convolution: the norm behind computes its own statistics)
            if ops.shadow_of(x) is None and gn is not None and _FUSE_GN_STATS and gn._sums_buf is not None and gn.num_groups == 32 and self.out_channels == 64:
                gn._stats_ready, sums = True, gn._sums_buf
            return ops.StemConvFn.apply(x, w, self._dw, sums, self._stem_hw)
        if self._prepadded:
            return F.conv2d(x, w, None, self.stride, 0, self.dilation, self.groups)
        k, s = self.kernel_size[0], self.stride[0]
        ih, iw = x.shape[-2:]
        ph = max((math.ceil(ih / s) - 1) * s + k - ih, 0)
        pw = max((math.ceil(iw / s) - 1) * s + k - iw, 0)
        if ph % 2 == 0 and pw % 2 == 0:   # symmetric SAME padding: let the convolution pad (no padded copy)
            return F.conv2d(x, w, None, self.stride, (ph // 2, pw // 2), self.dilation, self.groups)
        x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])
        return F.conv2d(x, w, None, self.stride, 0, self.dilation, self.groups)


class GroupNormAct(nn.GroupNorm):
    """resnetv2.py:35-49"""

    _direct_grad = False  # set by ResNetV2: the kernels accumulate dgamma/dbeta straight into .grad
    _sums_buf = None      # set by ResNetV2 per pass: pre-zeroed (N,32,2) f64 / (N,C,2) f32 scratch slices
    _ab_buf = None
    _sync_buf = None      # ... and the N * GN_SYNC_WORDS zero words of the one-pass backward (maed_groupnorm_bwd frame_sync: arrival counter + group sums per frame)
    _stats_ready = False  # set by the convolution in front when its epilogue filled _sums_buf (consumed by the next forward)

    def __init__(self, num_channels, num_groups=32, eps=1e-5, affine=True, apply_act=True):
        super().__init__(num_groups, num_channels, eps=eps, affine=affine)
        self.apply_act = apply_act

    def forward(self, x, residual=None, relu=None, lazy_res=False):
        """y = act(GN(x) [+ residual]); relu defaults to the layer's own activation flag"""
        relu = self.apply_act if relu is None else relu
        if ops.on_library_device(x) and self.num_groups == 32:
            ready, self._stats_ready = self._stats_ready and self._sums_buf is not None, False
            return ops.GroupNormFn.apply(x, residual, self.weight, self.bias, self.eps, relu, self._direct_grad, self._sums_buf, self._ab_buf, ready, lazy_res,
                                         self._sync_buf)
        x = F.group_norm(x, self.num_groups, self.weight.to(x.dtype), self.bias.to(x.dtype), self.eps)
        if residual is not None:
            x = x + residual
        return F.relu(x) if relu else x


class MaxPool2dSame(nn.Module):
    """resnetv2.py:61-72: pads with -inf"""

    def __init__(self, kernel_size=3, stride=2):
        super().__init__()
        self.kernel_size, self.stride = kernel_size, stride

    def forward(self, x):
        if ops.on_library_device(x) and self.kernel_size == 3 and self.stride == 2 and x.shape[1] % 8 == 0 and x.dtype in (torch.float32, torch.bfloat16):
            return ops.MaxPool3s2SameFn.apply(x)
        return F.max_pool2d(_same_pad(x, self.kernel_size, self.stride, value=-float("inf")), self.kernel_size, self.stride, 0)


class DownsampleConv(nn.Module):
    """resnetv2.py:207-216 (preact=False: conv + norm without activation)"""

    def __init__(self, in_chs, out_chs, stride=1):
        super().__init__()
        self.conv = StdConv2dSame(in_chs, out_chs, 1, stride=stride)
        self.norm = GroupNormAct(out_chs, apply_act=False)

    def forward(self, x):
        return self.norm(self.conv(x, gn=self.norm))


class Bottleneck(nn.Module):
    """resnetv2.py:159-204 non-pre-activation bottleneck; mid = out/4"""

    def __init__(self, in_chs, out_chs, stride=1, downsample=False):
        super().__init__()
        mid = out_chs // 4
        self.downsample = DownsampleConv(in_chs, out_chs, stride) if downsample else None
        self.conv1 = StdConv2dSame(in_chs, mid, 1)
        self.norm1 = GroupNormAct(mid)
        self.conv2 = StdConv2dSame(mid, mid, 3, stride=stride)
        self.norm2 = GroupNormAct(mid)
        self.conv3 = StdConv2dSame(mid, out_chs, 1)
        self.norm3 = GroupNormAct(out_chs, apply_act=False)
        self.drop_path = nn.Identity()

    def forward(self, x):
        if self.conv1._w_t is not None and torch.is_grad_enabled() and x.requires_grad:
            # x feeds conv1 AND the shortcut (identity or downsample): conv1 (a GEMM convolution) hands out an alias of x for
            # the shortcut, so the shortcut branch's gradient is added inside conv1's input-gradient GEMM epilogue instead of
            # by a separate autograd accumulation kernel
            # identity blocks: the shortcut's gradient is dy of norm3 masked by its ReLU bits -- never materialised: norm3's backward hands dy on,
            # conv1's input-gradient GEMM applies the bits while it adds (ops.GroupNormFn lazy_res / Conv1x1Fn lazy_short)
            lazy = self.downsample is None and _LAZY_RES_GRAD
            y, xa = self.conv1(x, fork=True, gn=self.norm1, lazy_short=lazy)
            shortcut = xa if self.downsample is None else self.downsample(xa)
            x = self.norm1(y)
            x = self.norm2(self.conv2(x, gn=self.norm2))
            return self.norm3(self.conv3(x, gn=self.norm3), residual=shortcut, relu=True, lazy_res=lazy)
        else:
            shortcut = x if self.downsample is None else self.downsample(x)
            x = self.norm1(self.conv1(x, gn=self.norm1))
        x = self.norm2(self.conv2(x, gn=self.norm2))      # (gn=: the convolution's epilogue accumulates the GroupNorm statistics)
        return self.norm3(self.conv3(x, gn=self.norm3), residual=shortcut, relu=True)   # GN + shortcut add + ReLU in one pass


class ResNetStage(nn.Module):
    """resnetv2.py:218-242"""

    def __init__(self, in_chs, out_chs, stride, depth):
        super().__init__()
        blocks = OrderedDict()
        for b in range(depth):
            blocks[str(b)] = Bottleneck(in_chs if b == 0 else out_chs, out_chs, stride=stride if b == 0 else 1, downsample=(b == 0))
        self.blocks = nn.Sequential(blocks)

    def forward(self, x):
        return self.blocks(x)


def _slots(module, **values):
    """per-pass hand-over slots of a layer (standardised weight, scratch slices): plain instance attributes, set without nn.Module.__setattr__'s parameter / buffer /
    submodule bookkeeping -- 750 assignments per train step, 2 us each through the Module (scripts/host_profile.py)"""
    module.__dict__.update(values)


class ResNetV2(nn.Module):
    """resnetv2.py:277-348 restricted to what the STE uses: preact=False, stem_type='same', no head."""

    def __init__(self, layers=(3, 4, 9), channels=(256, 512, 1024), in_chans=3, stem_chs=64, compute_dtype=torch.float32, f32_matmul=None, **_):
        super().__init__()
        self.compute_dtype = compute_dtype
        # compute_dtype = float32 only: this module's own engine for its fp32 matrix products -- None (follow the process-wide mode,
        # ops.set_float32_matmul_precision), "bf16x3" or "bf16x6".  Why a backbone may want more than the rest of the model: its 52 GroupNorms after
        # weight-standardised convolutions make the gradients of a freshly initialised network ill-conditioned (the fp32 reference arithmetic itself sits
        # 1.5e-2 from fp64 there, bf16x3 7e-2, bf16x6 1.6e-2: scripts/x3_probe.py, profiles/r03_x3_probe.txt)
        assert f32_matmul in (None, "bf16x3", "bf16x6"), f32_matmul
        self.f32_matmul = f32_matmul
        self.stem = nn.Sequential(OrderedDict([
            ("conv", StdConv2dSame(in_chans, stem_chs, 7, stride=2)),
            ("norm", GroupNormAct(stem_chs)),
            ("pool", MaxPool2dSame(3, 2)),
        ]))
        stages = OrderedDict()
        prev = stem_chs
        for i, (d, c) in enumerate(zip(layers, channels)):
            stages[str(i)] = ResNetStage(prev, c, stride=1 if i == 0 else 2, depth=d)
            prev = c
        self.stages = nn.Sequential(stages)
        self.num_features = prev
        self.norm = nn.Identity()
        self.head = nn.Identity()
        for m in self.modules():  # resnetv2.py:330-335
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self._convs = [m for m in self.modules() if isinstance(m, StdConv2dSame)]
        self._norms = [m for m in self.modules() if isinstance(m, GroupNormAct)]
        for m in self._norms:
            m._direct_grad = True
        # 1x1 convolutions (stride 1, and the stride-2 downsample shortcuts on packed pixels) with GEMM-friendly channel counts run on
        # libmaed_hip in bf16 mode (ops.Conv1x1Fn)
        strides = ((1, 1), (2, 2)) if os.environ.get("MAED_CONV1X1_S2", "1") == "1" else ((1, 1),)     # (A/B knob: stride-2 ones on MIOpen)
        self._gemm_convs = [i for i, c in enumerate(self._convs) if c.kernel_size == (1, 1) and c.stride in strides
                            and c.in_channels % 64 == 0 and c.out_channels % 64 == 0]
        if os.environ.get("MAED_GEMM_CONVS", "1") == "0":      # measurement knob: every convolution on MIOpen
            self._gemm_convs = []
        # MAED_CONV3X3=own: the stride-1 3x3 convolutions hand their weights / weight gradients over like the GEMM convolutions
        # (round 4: the two stride-2 ones too -- their input gradient runs on maed_conv3x3_s2_dgrad, which reads the transposed image; MAED_CONV3X3_S2=0: as before)
        s3 = ((1, 1), (2, 2)) if os.environ.get("MAED_CONV3X3_S2", "1") == "1" else ((1, 1),)
        self._own3x3 = [i for i, c in enumerate(self._convs) if _OWN_CONV3X3 and c.kernel_size == (3, 3) and c.stride in s3
                        and c.in_channels % 64 == 0 and c.out_channels % 64 == 0]
        # the stem (7x7 stride 2, 3 -> 64) on maed_stem7x7s2_* in bf16 mode when the frame geometry allows (decided per forward); MAED_STEM_OWN=0: vendor convolution
        self.stem.conv._gn_behind = (self.stem.norm,)
        self._own_stem_now = []
        self._w_std_t, self._dw_slices, self._dw_arena = {}, {}, None
        self._pending_backwards = 0
        self._grad_forward_seen = False     # a grad-enabled forward ran since the bucketer last looked (ddp.GradBucketer.finish)
        self.grads_ready = None  # callback(self) set by the data-parallel gradient bucketer
        # per-stage groups (MAED_WS_PER_STAGE): the stem rides with stage 0
        conv_pos = {id(c): i for i, c in enumerate(self._convs)}
        parts = [[self.stem, self.stages[0]]] + [[st] for st in list(self.stages)[1:]]
        self._ws_groups = [_WsGroup(self, [conv_pos[id(m)] for part in ps for m in part.modules() if isinstance(m, StdConv2dSame)],
                                    [m for part in ps for m in part.modules() if isinstance(m, GroupNormAct)]) for ps in parts]

    def conv_weights(self):
        return [c.weight for c in self._convs]

    @property
    def _direct_convs(self):
        """convolutions on the library's own kernels (transposed weight image + fp32 dW slice from WeightStdFn); computed on access so that
        switching `_gemm_convs` off at run time (tests/test_gpu_model.py's all-MIOpen variant) keeps its meaning"""
        return list(self._gemm_convs) + list(self._own3x3) + list(self._own_stem_now)

    def fused_parameters(self):
        """parameters whose gradients the HIP kernels write directly into .grad: conv weights (batched
        weight-standardisation backward, the LAST kernel of the backbone backward) and GroupNorm affine parameters"""
        return self.conv_weights() + [t for m in self._norms for t in (m.weight, m.bias)]

    def forward_features(self, x):
        if not ops.on_library_device(x):
            return self.stages(self.stem(x))
        # every per-pass hand-over slot (pre-padded stem input, standardised weights, GroupNorm scratch) is set INSIDE the try: an exception anywhere -- alignment,
        # out of memory, an unsupported size -- must not leave the stem marked "pre-padded" (the next forward would convolve an unpadded image: silently wrong)
        try:
            y = self._forward_features_library(x)
            ops.shadow_keep_only(y)          # twin mode: the layers' fp32 activations are dead now; only the output's shadow has a consumer (HybridEmbed's projection)
            return y
        except BaseException:
            ops.shadow_clear()
            raise
        finally:
            _slots(self.stem.conv, _prepadded=False, _stem_hw=None)
            self._own_stem_now = []
            for c in self._convs:
                _slots(c, _w_std=None, _w_t=None, _dw=None, _prec=None)
            for m in self._norms:
                _slots(m, _sums_buf=None, _ab_buf=None, _sync_buf=None)

    def _forward_features_library(self, x):
        stem = self.stem.conv
        cdt = self.compute_dtype
        stem_fused = (x.dtype == torch.float32 and x.is_contiguous() and x.shape[1] <= 4 and not x.requires_grad and os.environ.get("MAED_STEM_INPUT", "1") == "1"
                      and stem.dilation == (1, 1) and stem.kernel_size[0] == stem.kernel_size[1])
        own_ok = (os.environ.get("MAED_STEM_OWN", "1") == "1" and stem.kernel_size == (7, 7) and stem.stride == (2, 2)
                  and stem.in_channels <= 3 and stem.out_channels == 64 and stem.groups == 1 and ops.stem7x7s2_supported(x.shape[2], x.shape[3], x.shape[0]))
        # "bf16x3 forward / bf16 backward from bf16 twins" (ops.set_float32_backward_precision("bf16")): the autograd graph of this pass is the bf16 mode's -- every
        # tensor autograd sees is a bf16 twin, the fp32 tensors the forward chain computes with travel beside them as shadows (ops.shadow_put).  Training passes of an
        # fp32 model only; anything the bf16 graph cannot express here (no fused stem input, a frame size the library's stem does not take) keeps the one-plane backward.
        # (every convolution must be one of the library's: a layer on the framework's convolution would produce a bf16 tensor without a shadow)
        twin = (cdt == torch.float32 and ops.bwd_twin() and torch.is_grad_enabled() and stem_fused and own_ok
                and len(set(self._gemm_convs) | set(self._own3x3) | {0}) == len(self._convs) and any(p.requires_grad for p in self.fused_parameters()))
        ops.shadow_clear()
        if twin:
            cdt = torch.bfloat16
            ops.TWIN_FORWARDS[0] += 1
        if stem_fused:
            # cast + channels_last + the stem's TF-SAME padding in one pass (the framework: three); the stem convolution below sees an already padded image
            own = cdt == torch.bfloat16 and own_ok
            self._own_stem_now = [0] if own else []
            _slots(stem, _prepadded="own" if own else True, _stem_hw=(x.shape[2], x.shape[3]))
            x32p = ops.stem_input(x, torch.float32, stem.kernel_size[0], stem.stride[0], own=False) if twin else None
            x = ops.stem_input(x, cdt, stem.kernel_size[0], stem.stride[0], own=own)
            if twin:
                ops.shadow_put(x, x32p)
        else:
            x = x.to(dtype=cdt, memory_format=torch.channels_last)
        ws = None if _ws_per_stage() else self._standardise(self, cdt, twin)
        # GroupNorm scratch for all layers of this pass: ONE zero-fill each instead of a memset per layer and direction
        N = x.shape[0]
        sums = torch.zeros(len(self._norms), N, 32, 2, dtype=torch.float64, device=x.device)
        ab = None
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.fused_parameters()):
            self._grad_forward_seen = True
        if torch.is_grad_enabled() and any(p.requires_grad for p in (self._norms[0].weight, self._convs[0].weight)):
            # (+ N * GN_SYNC_WORDS zero words per layer behind the partial sums: per-frame arrival counter + group sums of the one-pass GroupNorm backward --
            #  same fill, single use)
            ab = torch.zeros(N * 2 * sum(m.num_channels for m in self._norms) + N * ops.GN_SYNC_WORDS * len(self._norms), dtype=torch.float32, device=x.device)
        if ws is not None:
            for i, (c, w) in enumerate(zip(self._convs, ws)):
                _slots(c, _w_std=w, _w_t=self._w_std_t.get(i), _dw=self._dw_slices.get(i), _prec=self.f32_matmul)
        off = 0
        for i, m in enumerate(self._norms):
            _slots(m, _sums_buf=sums[i])
            if ab is not None:
                n = N * 2 * m.num_channels
                _slots(m, _ab_buf=ab[off:off + n].view(N, m.num_channels, 2), _sync_buf=ab[off + n:off + n + N * ops.GN_SYNC_WORDS])
                off += n + N * ops.GN_SYNC_WORDS
        if ws is not None:
            return self.stages(self.stem(x))
        for gi, g in enumerate(self._ws_groups):       # standardise a stage's weights right before it runs: its autograd node
            wg = self._standardise(g, cdt, twin)                                                       # then fires right after its backward
            for k, (ci, w) in enumerate(zip(g.conv_idx, wg)):
                c = self._convs[ci]
                _slots(c, _w_std=w, _w_t=g._w_std_t.get(k), _dw=g._dw_slices.get(k), _prec=self.f32_matmul)
            x = self.stages[0](self.stem(x)) if gi == 0 else self.stages[gi](x)
        # backward order is last stage first: every group but the one that runs last may finish on the side stream (ops.WeightStdFn.backward)
        runs = [g for g in self._ws_groups if g._pending_backwards > 0]
        for k, g in enumerate(self._ws_groups):
            g._ws_on_side = bool(runs) and g is not runs[0] and g in runs
        return x

    def _standardise(self, owner, cdt, twin):
        """the batched weight standardisation of `owner` (the backbone or one stage group) in the pass's dtype; twin: + the fp32 images of the same weights for the
        forward products, registered as the shadows of the bf16 ones (no autograd node, no transposed images, no gradient slices: the backward is the bf16 graph's)"""
        eps = self._convs[0].eps
        ws = ops.WeightStdFn.apply(owner, cdt, eps, *owner.conv_weights())
        if twin:
            import types
            with torch.no_grad():
                ws32 = ops.WeightStdFn.apply(types.SimpleNamespace(_direct_convs=[], f32_matmul=self.f32_matmul, _pending_backwards=0), torch.float32, eps,
                                             *[w.detach() for w in owner.conv_weights()])
            for w16, w32 in zip(ws, ws32):
                ops.shadow_put(w16, w32)
        return ws

    def forward(self, x, seqlen=8):
        return self.forward_features(x)
