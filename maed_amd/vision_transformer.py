"""Spatial-Temporal Encoder (reference: lib/models/vision_transformer.py).

Same classes, constructor arguments, attribute names and state_dict keys as the reference
(`Mlp`, `Attention`, `Block`, `HybridEmbed`, `VisionTransformer`, `vit_custom_resnet50_224_in21k`),
so reference checkpoints and train.py/eval.py-style drivers drop in.  Underneath, a Block is ONE
call into libmaed_hip.so per direction (ops.STEBlockFn -> maed_ste_block_fwd/bwd): LayerNorm, qkv
GEMM, temporal + spatial attention reading qkv in place, attentive addition, proj and MLP GEMMs
with fused bias/GELU/residual epilogues.  The residual stream is fp32; activations and GEMM
operands are in `compute_dtype` (torch.bfloat16 for throughput, torch.float32 for parity).

Extra keyword arguments relative to the reference (defaults = the reference's hard-coded values,
SURVEY.md section 0): `embed_dim`, `max_seqlen`, `img_size`, `compute_dtype`, `impl`.
st_mode='parallel' (the configured mode, configs/config_stage2.yaml:75) is the fused path; the ablation modes 'series',
'vanilla', 'temporal', 'coupling' are composed from the same kernels in maed_amd/ste_modes.py.
"""
from collections import OrderedDict
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import ops
from . import ste_modes
from .resnetv2 import ResNetV2

ST_MODES = ('parallel', 'series', 'vanilla', 'temporal', 'coupling')


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    """vision_transformer.py:39-93 (same distribution; torch's implementation)."""
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


class Mlp(nn.Module):
    """vision_transformer.py:96-112: fc2(GELU_erf(fc1(x))); dropouts are rate 0 on this path."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)
        self._cache = ops.WeightCache()

    def forward(self, x, compute_dtype=torch.float32):
        """Stand-alone (inference) use; inside a Block the MLP runs fused in maed_ste_block_fwd."""
        shp = x.shape
        if _needs_grad(x, self.fc1.weight):     # stand-alone differentiable use: the staged Function (inside a Block the MLP runs fused)
            y = ste_modes.MlpFn.apply(x.reshape(-1, shp[-1]).to(compute_dtype), self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, self._cache)
            return y.reshape(*shp[:-1], -1)
        (w1, _), (w2, _) = self._cache.get([self.fc1.weight, self.fc2.weight], compute_dtype)
        a = x.reshape(-1, shp[-1]).to(compute_dtype)
        act, _ = ops.gemm_nt(a, w1, L.EPI_GELU, bias=self.fc1.bias)
        y = ops.gemm_nt(act, w2, L.EPI_STORE_F32, bias=self.fc2.bias)
        return y.reshape(*shp[:-1], -1)


class Attention(nn.Module):
    """vision_transformer.py:115-240.  st_mode='parallel' (:146-158,176) is what Block fuses; the other modes
    (:139-145,160-173) run through ste_modes.attention."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., st_mode='vanilla'):
        super().__init__()
        if st_mode not in ST_MODES:
            raise NotImplementedError(st_mode)       # the reference raises at forward time (:175)
        if dim != 64 * num_heads:
            raise NotImplementedError(f"head dim must be 64 (dim={dim}, heads={num_heads})")
        if qk_scale is not None or attn_drop or proj_drop:
            raise NotImplementedError("qk_scale / attention dropout are not used on the MAED path")
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.proj = nn.Linear(dim, dim)
        self.mode = st_mode
        if st_mode == 'parallel':                    # the only mode with the attentive-addition weights (:126-128)
            self.ts_attn = nn.Linear(dim * 2, dim * 2)
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj_drop = nn.Dropout(proj_drop)
        self._cache = ops.WeightCache()

    def forward(self, x, seqlen=1, compute_dtype=torch.float32, impl=L.IMPL_AUTO, return_parts=False):
        """Stand-alone use.  'parallel': inference only (inside a Block it runs fused in maed_ste_block_fwd);
        the other modes are differentiable here."""
        if self.mode != 'parallel':
            return ste_modes.attention(self, x.to(compute_dtype), seqlen, compute_dtype, impl)
        if _needs_grad(x, self.qkv.weight) and not return_parts:   # stand-alone differentiable use (inside a Block: one fused call per direction)
            return ste_modes.attention_parallel(self, x, seqlen, compute_dtype, impl)
        Fr, P, C_ = x.shape
        (wq, _), (wt, _), (wp, _) = self._cache.get([self.qkv.weight, self.ts_attn.weight, self.proj.weight], compute_dtype)
        a = x.reshape(-1, C_).to(compute_dtype)
        qkv = ops.gemm_nt(a, wq, L.EPI_STORE, bias=self.qkv.bias).view(Fr, P, 3 * C_)
        x_t, _ = ops.attn_temporal_fwd(qkv, self.num_heads, seqlen)
        x_s, _ = ops.attn_spatial_fwd(qkv, self.num_heads, impl)
        means = ops.st_colmean(x_s, x_t)
        logits = ops.gemm_nt(means, wt, L.EPI_STORE_F32, bias=self.ts_attn.bias)
        mix = ops.st_mix_fwd(x_s, x_t, logits)
        out = ops.gemm_nt(mix.view(-1, C_), wp, L.EPI_STORE_F32, bias=self.proj.bias).view(Fr, P, C_)
        if return_parts:
            return out, dict(qkv=qkv, x_s=x_s, x_t=x_t, logits=logits, mix=mix)
        return out


_LIN_FIELDS = [("w_" + n, "wt_" + n, "b_" + n) for n in ("qkv", "ts", "proj", "fc1", "fc2")]      # maed_block_params / maed_block_grads field names per Linear


class Block(nn.Module):
    """vision_transformer.py:244-261.  forward = one fused call (ops.STEBlockFn) in 'parallel' mode, the staged
    composition of ste_modes.block otherwise."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, st_mode='vanilla',
                 compute_dtype=torch.bfloat16, impl=L.IMPL_AUTO):
        super().__init__()
        if drop or drop_path:
            raise NotImplementedError("dropout / stochastic depth are rate 0 on the MAED path (vision_transformer.py:568)")
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop, st_mode=st_mode)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.dim, self.num_heads, self.hidden = dim, num_heads, int(dim * mlp_ratio)
        self.compute_dtype, self.impl = compute_dtype, impl
        self._cache = ops.WeightCache()
        self._pending_backwards = 0
        self.grads_ready = None  # callback(block) set by the data-parallel gradient bucketer
        self.fused = st_mode == 'parallel'

    # ---- C structs for the fused kernels -------------------------------------------------------
    def _linears(self):
        return [self.attn.qkv, self.attn.ts_attn, self.attn.proj, self.mlp.fc1, self.mlp.fc2]

    def _hot(self):
        """(linears, LayerNorm parameters, fused parameter list) looked up once: the attribute chains below go through nn.Module.__getattr__ (2-3 us a hop), twelve
        times per train step and block.  Parameters are replaced in place by load_state_dict / optimizers; a module surgery that swaps Parameter OBJECTS must
        drop `_hot_cache` (ParamArena re-points .data, which keeps the objects)."""
        c = self.__dict__.get("_hot_cache")
        if c is None:
            lin = self._linears()
            ln = (self.norm1.weight, self.norm1.bias, self.norm2.weight, self.norm2.bias)
            c = (lin, [l.weight for l in lin], [l.bias for l in lin], ln, list(ln) + [t for l in lin for t in (l.weight, l.bias) if t is not None])
            self.__dict__["_hot_cache"] = c
        return c

    def _c_params_masters(self):
        """parameters of a FORWARD on the fp32 masters themselves (no images: the forward reads w_* only) -- the twin forward of ops.STEBlockFn"""
        lin, weights, biases, ln, _ = self._hot()
        p = L.BlockParams()
        p.ln1_g, p.ln1_b, p.ln2_g, p.ln2_b = ln[0].data_ptr(), ln[1].data_ptr(), ln[2].data_ptr(), ln[3].data_ptr()
        for name, b, w in zip(_LIN_FIELDS, biases, weights):
            assert w.is_contiguous()
            setattr(p, name[0], w.data_ptr())
            setattr(p, name[2], b.data_ptr() if b is not None else None)
        return p

    def _c_params(self, dtype, backward=False):
        lin, weights, biases, ln, _ = self._hot()
        # (a backward in another dtype than the forward's -- bf16 twins behind an fp32 forward -- has its own image cache: one cache holds one dtype)
        cache = self._cache if (not backward or dtype == self.compute_dtype) else self.__dict__.setdefault("_cache_bwd", ops.WeightCache())
        w = cache.get(weights, dtype)
        self.__dict__["_keep"] = w
        p = L.BlockParams()
        p.ln1_g, p.ln1_b, p.ln2_g, p.ln2_b = ln[0].data_ptr(), ln[1].data_ptr(), ln[2].data_ptr(), ln[3].data_ptr()
        for name, b, (wc, wt) in zip(_LIN_FIELDS, biases, w):
            setattr(p, name[0], wc.data_ptr())
            setattr(p, name[1], wt.data_ptr())
            setattr(p, name[2], b.data_ptr() if b is not None else None)
        return p

    def _c_grads(self):
        g = L.BlockGrads()

        def grad_ptr(prm):
            if prm is None:
                return None
            if prm.grad is None:
                prm.grad = torch.zeros_like(prm)
            return prm.grad.data_ptr()

        _, weights, biases, ln, _ = self._hot()
        g.ln1_g, g.ln1_b, g.ln2_g, g.ln2_b = grad_ptr(ln[0]), grad_ptr(ln[1]), grad_ptr(ln[2]), grad_ptr(ln[3])
        for name, wgt, b in zip(_LIN_FIELDS, weights, biases):
            setattr(g, name[0], grad_ptr(wgt))
            setattr(g, name[2], grad_ptr(b))
        return g

    def fused_parameters(self):
        """parameters whose gradients the fused backward writes straight into .grad (none in the staged modes: autograd owns them)"""
        if not self.fused:
            return []
        return self._hot()[4]

    def forward(self, x, seqlen=1):
        if not self.fused:
            return ste_modes.block(self, x, seqlen)
        Fr, P, C_ = x.shape
        dims = (Fr, P, C_, self.num_heads, seqlen, self.hidden, ops.dt_code(self.compute_dtype), self.impl, self.norm1.eps)
        return ops.STEBlockFn.apply(x.float(), self, dims, self.num_heads, seqlen, *self.fused_parameters())


class HybridEmbed(nn.Module):
    """vision_transformer.py:287-311: CNN feature map -> 1x1 projection -> (F, HW, C) tokens.
    The projection is a GEMM on the channels_last feature map viewed as (F*HW, Cin): no transpose copy."""

    def __init__(self, backbone, img_size=224, feature_size=None, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.backbone = backbone
        with torch.no_grad():  # probe the output grid the way the reference does (:296-298)
            training = backbone.training
            backbone.eval()
            o = backbone(torch.zeros(1, in_chans, img_size, img_size))
            backbone.train(training)
        self.num_patches = o.shape[-2] * o.shape[-1]
        self.proj = nn.Conv2d(o.shape[1], embed_dim, 1)
        self._cache = ops.WeightCache()

    def forward(self, x):
        f = self.backbone(x)                                      # (F, Cin, h, w) channels_last on GPU
        Fr, Cin, h, w = f.shape
        a = f.permute(0, 2, 3, 1).reshape(Fr * h * w, Cin)         # a view when f is channels_last
        f32 = ops.shadow_of(f)                                     # "bf16" backward mode: f is the bf16 twin autograd sees, its fp32 shadow is the operand
        if f32 is not None:
            ops.shadow_put(a, f32.permute(0, 2, 3, 1).reshape(Fr * h * w, Cin))
        try:
            y = ops.LinearFn.apply(a, self.proj.weight.view(self.proj.out_channels, Cin), self.proj.bias, self._cache, self.proj.weight, self.proj.bias)
        finally:
            ops.shadow_clear()                                     # the backbone's fp32 activations were transient: everything behind the projection is fp32 anyway
        return y.view(Fr, h * w, -1)


class VisionTransformer(nn.Module):
    """vision_transformer.py:314-413 (hybrid input stage)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=False, qk_scale=None, representation_size=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., hybrid_backbone=None, norm_layer=nn.LayerNorm,
                 st_mode='vanilla', max_seqlen=16, compute_dtype=torch.bfloat16, impl=L.IMPL_AUTO):
        super().__init__()
        if hybrid_backbone is None:
            raise NotImplementedError("only the hybrid (CNN feature map) input stage is on the MAED path")
        if drop_rate or attn_drop_rate or drop_path_rate:
            raise NotImplementedError("dropout rates are 0 on the MAED path (vision_transformer.py:568)")
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.compute_dtype = compute_dtype
        self.patch_embed = HybridEmbed(hybrid_backbone, img_size=img_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  norm_layer=norm_layer, st_mode=st_mode, compute_dtype=compute_dtype, impl=impl)
            for _ in range(depth)])
        for i, blk in enumerate(self.blocks):
            blk._chain_index = i        # position in the chain: ops.STEBlockFn's residual-gradient hand-off knows which block a copy was left for
        self.norm = norm_layer(embed_dim)
        self.st_mode = st_mode
        if representation_size:
            self.num_features = representation_size
            self.pre_logits = nn.Sequential(OrderedDict([('fc', nn.Linear(embed_dim, representation_size)), ('act', nn.Tanh())]))
        else:
            self.pre_logits = nn.Identity()
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        trunc_normal_(self.pos_embed, std=.02)
        trunc_normal_(self.cls_token, std=.02)
        if st_mode in ('coupling', 'parallel', 'series'):                             # :363-365
            self.temp_embed = nn.Parameter(torch.zeros(1, max_seqlen, 1, embed_dim))  # reference: 16 slots (:364)
            trunc_normal_(self.temp_embed, std=.02)
        else:   # 'vanilla' / 'temporal' add no temporal embedding (:396): the embed kernel gets one all-zero slot
            self.register_buffer('_no_temp_embed', torch.zeros(1, 1, 1, embed_dim), persistent=False)
        self.apply(self._init_weights)
        self._cache = ops.WeightCache()

    def _init_weights(self, m):  # vision_transformer.py:368-375
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward_tokens(self, x, seqlen=1):
        """everything up to (and including) the last Block: (F,3,S,S) -> fp32 tokens (F,P,C)"""
        patch = self.patch_embed(x)
        if hasattr(self, 'temp_embed'):
            if seqlen > self.temp_embed.shape[1]:
                raise ValueError(f"seqlen={seqlen} exceeds max_seqlen={self.temp_embed.shape[1]}")
            tok = ops.EmbedAddFn.apply(patch, self.cls_token, self.pos_embed, self.temp_embed, seqlen)
        else:
            tok = ops.EmbedAddFn.apply(patch, self.cls_token, self.pos_embed, self._no_temp_embed, 1)
        twins = ops.TWIN_FORWARDS[0]
        for blk in self.blocks:
            tok = blk(tok, seqlen)
        if ops.TWIN_FORWARDS[0] != twins:
            # twin forwards leave cast passes on a library stream that write the blocks' saved arenas: joined here, once per chain, so that a graph that is dropped
            # without a backward (or an exception further down) can never hand an arena back to the allocator while a cast still writes it (costs the
            # overlap of the LAST block's cast pass only)
            ops.twin_join()
        return tok

    def forward_features(self, x, seqlen=1):
        tok = self.forward_tokens(x, seqlen)
        Fr, P, C_ = tok.shape
        fc = self.pre_logits.fc if isinstance(self.pre_logits, nn.Sequential) else None
        if _needs_grad(tok, self.norm.weight):
            # tail of the training graph ((F,C) cls rows): the same library kernels behind autograd Functions (LayerNorm fwd/bwd, GEMM with
            # the tanh epilogue, maed_tanh_bwd, weight-gradient GEMMs); autograd's slice backward puts the row gradients back into (F,P,C)
            y = ste_modes.LayerNormFn.apply(tok[:, 0], self.norm.weight, self.norm.bias, self.norm.eps, self.compute_dtype if fc is not None else torch.float32, True)
            return ste_modes.TanhLinearFn.apply(y, fc.weight, fc.bias, self._cache, True).float() if fc is not None else y
        # inference: LayerNorm only the cls rows (row stride P*C), pre_logits GEMM with fused tanh
        y, _, _ = ops.layernorm_fwd(tok, self.norm.weight, self.norm.bias, self.compute_dtype if fc is not None else torch.float32,
                                    eps=self.norm.eps, row_stride=P * C_, rows=Fr)
        if fc is None:
            return y
        (w, _), = self._cache.get([fc.weight], self.compute_dtype)
        return ops.gemm_nt(y, w, L.EPI_TANH, bias=fc.bias).float()

    def forward(self, x, seqlen=1):
        return self.head(self.forward_features(x, seqlen))


def vit_custom_resnet50_224_in21k(num_blocks, num_heads, st_mode, pretrained=False, embed_dim=768, img_size=224,
                                  max_seqlen=16, compute_dtype=torch.bfloat16, impl=L.IMPL_AUTO, **kwargs):
    """vision_transformer.py:560-576.  `pretrained` downloads nothing here (no egress): load the
    reference's jx_vit_base_resnet50_224_in21k state_dict yourself with load_state_dict(strict=False)."""
    if pretrained:
        raise NotImplementedError("pretrained=True needs network access; load the checkpoint with load_state_dict")
    backbone = ResNetV2(layers=(3, 4, 9), in_chans=kwargs.get('in_chans', 3), compute_dtype=compute_dtype)
    kwargs.pop('num_classes', None)
    return VisionTransformer(img_size=img_size, patch_size=16, embed_dim=embed_dim, depth=num_blocks, num_heads=num_heads,
                             hybrid_backbone=backbone, mlp_ratio=4, qkv_bias=True, representation_size=embed_dim,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), st_mode=st_mode, num_classes=-1,
                             max_seqlen=max_seqlen, compute_dtype=compute_dtype, impl=impl, **kwargs)
