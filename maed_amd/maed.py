"""Top-level model (reference: lib/models/maed.py).

    MAED(encoder='ste', num_blocks=6, num_heads=12, st_mode='parallel', decoder='ktd', hidden_dim=1024, **kwargs)
    forward(x (N,T,3,H,W), J_regressor=None) -> {'theta','verts','kp_2d','kp_3d','rotmat'}   (maed.py:52-67)
    extract_feature(x) -> (N,T,feat)                                                              (maed.py:43-50)

Extra keyword arguments (defaults = the reference's hard-coded values): embed_dim=768, max_seqlen=16,
img_size=224, compute_dtype=torch.bfloat16 (torch.float32 = parity mode), impl, smpl_arrays
(dict with the SMPL model arrays; None -> deterministic synthetic stand-in, see maed_amd/smpl.py).
Only encoder='ste' / decoder='ktd' (the configured pair, configs/config_stage2.yaml:70-78) exist.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from .ktd import KTD
from .vision_transformer import vit_custom_resnet50_224_in21k


class _TrainingTail(nn.Module):
    """final LayerNorm on the cls rows + pre_logits + KTD + SMPL + projection: the (F, .)-row ATen part of the
    training graph (~1500 tiny launches per step), isolated so it can be captured into ONE hipGraph."""

    def __init__(self, encoder, decoder):
        super().__init__()
        self.norm, self.pre_logits, self.decoder = encoder.norm, encoder.pre_logits, decoder

    def forward(self, cls_tok):
        y = F.layer_norm(cls_tok, (cls_tok.shape[-1],), self.norm.weight, self.norm.bias, self.norm.eps)
        if isinstance(self.pre_logits, nn.Sequential):
            y = torch.tanh(F.linear(y, self.pre_logits.fc.weight, self.pre_logits.fc.bias))
        o = self.decoder(y, seqlen=1)
        return o['theta'], o['verts'], o['kp_2d'], o['kp_3d'], o['rotmat']


class MAED(nn.Module):
    def __init__(self, encoder='ste', num_blocks=6, num_heads=12, st_mode='parallel', decoder='ktd', hidden_dim=1024,
                 embed_dim=768, max_seqlen=16, img_size=224, compute_dtype=torch.bfloat16, impl=L.IMPL_AUTO,
                 smpl_arrays=None, **kwargs):
        super().__init__()
        self.encoder_type = encoder
        if encoder.lower() != 'ste':
            raise NotImplementedError(encoder)       # maed.py:41 ('cnn' = stage-1 torchvision ResNet-50: out of scope)
        self.encoder = vit_custom_resnet50_224_in21k(num_blocks, num_heads, st_mode, embed_dim=embed_dim, img_size=img_size,
                                                     max_seqlen=max_seqlen, compute_dtype=compute_dtype, impl=impl)
        self.decoder_type = decoder
        if decoder.lower() != 'ktd':
            raise NotImplementedError(decoder)       # maed.py:29 ('iterative' SPIN regressor: SURVEY 8(f) rank 3)
        self.decoder = KTD(feat_dim=self.encoder.num_features, hidden_dim=hidden_dim, smpl_arrays=smpl_arrays)
        self._graphed_tail, self._graphed_frames = None, 0

    def graph_training_tail(self, n_frames):
        """Capture forward AND backward of the decoder tail for a fixed number of frames into hipGraphs
        (torch.cuda.make_graphed_callables): launch-bound work replayed as two graph launches per step.
        Call AFTER parameters have reached their final storage (e.g. after ddp.ParamArena) and in train mode."""
        tail = _TrainingTail(self.encoder, self.decoder)
        dev = self.encoder.norm.weight.device
        sample = torch.randn(n_frames, self.encoder.embed_dim, device=dev, requires_grad=True)
        self._graphed_tail = torch.cuda.make_graphed_callables(tail, (sample,))
        self._graphed_frames = n_frames
        return self

    def extract_feature(self, x):
        batch_size, seqlen = x.shape[:2]
        x = x.reshape(-1, x.shape[-3], x.shape[-2], x.shape[-1])
        return self.encoder(x).reshape(batch_size, seqlen, -1)  # note: the reference omits seqlen here too (maed.py:47)

    def forward(self, x, J_regressor=None, **kwargs):
        batch_size, seqlen = x.shape[:2]
        x = x.reshape(-1, x.shape[-3], x.shape[-2], x.shape[-1])
        use_graph = (self._graphed_tail is not None and self.training and torch.is_grad_enabled() and J_regressor is None
                     and not kwargs and x.shape[0] == self._graphed_frames)
        if use_graph:
            tok = self.encoder.forward_tokens(x, seqlen)
            theta, verts, kp_2d, kp_3d, rotmat = self._graphed_tail(tok[:, 0].contiguous())
            output = dict(theta=theta, verts=verts, kp_2d=kp_2d, kp_3d=kp_3d, rotmat=rotmat)
        else:
            xf = self.encoder(x, seqlen=seqlen)
            output = self.decoder(xf, seqlen=seqlen, J_regressor=J_regressor, **kwargs)
        output['theta'] = output['theta'].reshape(batch_size, seqlen, -1)
        output['verts'] = output['verts'].reshape(batch_size, seqlen, -1, 3)
        output['kp_2d'] = output['kp_2d'].reshape(batch_size, seqlen, -1, 2)
        output['kp_3d'] = output['kp_3d'].reshape(batch_size, seqlen, -1, 3)
        output['rotmat'] = output['rotmat'].reshape(batch_size, seqlen, -1, 3, 3)
        return output
