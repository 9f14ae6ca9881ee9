"""Top-level model (reference: lib/models/maed.py).

    MAED(encoder='ste', num_blocks=6, num_heads=12, st_mode='parallel', decoder='ktd', hidden_dim=1024, **kwargs)
    forward(x (N,T,3,H,W), J_regressor=None) -> {'theta','verts','kp_2d','kp_3d','rotmat'}   (maed.py:52-67)
    extract_feature(x) -> (N,T,feat)                                                              (maed.py:43-50)

Extra keyword arguments (defaults = the reference's hard-coded values): embed_dim=768, max_seqlen=16,
img_size=224, compute_dtype=torch.bfloat16 (torch.float32 = parity mode), impl, smpl_arrays
(dict with the SMPL model arrays; None -> deterministic synthetic stand-in, see maed_amd/smpl.py).
encoder='ste' only ('cnn' is the stage-1 torchvision ResNet-50); decoder 'ktd' (configured, config_stage2.yaml:70-78)
or 'iterative' (spin.py Regressor; extra kwarg smpl_mean_params).
"""
import torch
import torch.nn as nn

from . import _lib as L
from .ktd import KTD
from .vision_transformer import vit_custom_resnet50_224_in21k


class MAED(nn.Module):
    def __init__(self, encoder='ste', num_blocks=6, num_heads=12, st_mode='parallel', decoder='ktd', hidden_dim=1024,
                 embed_dim=768, max_seqlen=16, img_size=224, compute_dtype=torch.bfloat16, impl=L.IMPL_AUTO,
                 smpl_arrays=None, backbone_f32_matmul=None, **kwargs):
        super().__init__()
        self.encoder_type = encoder
        if encoder.lower() != 'ste':
            raise NotImplementedError(encoder)       # maed.py:41 ('cnn' = stage-1 torchvision ResNet-50: out of scope)
        self.encoder = vit_custom_resnet50_224_in21k(num_blocks, num_heads, st_mode, embed_dim=embed_dim, img_size=img_size,
                                                     max_seqlen=max_seqlen, compute_dtype=compute_dtype, impl=impl)
        # compute_dtype = float32: the backbone's own fp32 matmul engine (resnetv2.ResNetV2.f32_matmul); None = the process-wide mode
        self.encoder.patch_embed.backbone.f32_matmul = backbone_f32_matmul
        self.decoder_type = decoder
        if decoder.lower() == 'ktd':                 # maed.py:24-29
            self.decoder = KTD(feat_dim=self.encoder.num_features, hidden_dim=hidden_dim, smpl_arrays=smpl_arrays)
            import os
            if compute_dtype == torch.bfloat16 and os.environ.get("MAED_HEAD_X3", "1") == "1":        # (A/B knob: 0 = the head on the exact fp32 VALU kernels)
                self.decoder.head_matmul = "bf16x3"
        elif decoder.lower() == 'iterative':
            from .iterative import Regressor
            self.decoder = Regressor(feat_dim=self.encoder.num_features, hidden_dim=hidden_dim, smpl_arrays=smpl_arrays,
                                     smpl_mean_params=kwargs.get('smpl_mean_params'))
        else:
            raise NotImplementedError(decoder)

    def extract_feature(self, x):
        batch_size, seqlen = x.shape[:2]
        x = x.reshape(-1, x.shape[-3], x.shape[-2], x.shape[-1])
        return self.encoder(x).reshape(batch_size, seqlen, -1)  # note: the reference omits seqlen here too (maed.py:47)

    def forward(self, x, J_regressor=None, **kwargs):
        batch_size, seqlen = x.shape[:2]
        x = x.reshape(-1, x.shape[-3], x.shape[-2], x.shape[-1])
        xf = self.encoder(x, seqlen=seqlen)
        output = self.decoder(xf, seqlen=seqlen, J_regressor=J_regressor, **kwargs)
        output['theta'] = output['theta'].reshape(batch_size, seqlen, -1)
        output['verts'] = output['verts'].reshape(batch_size, seqlen, -1, 3)
        output['kp_2d'] = output['kp_2d'].reshape(batch_size, seqlen, -1, 2)
        output['kp_3d'] = output['kp_3d'].reshape(batch_size, seqlen, -1, 3)
        output['rotmat'] = output['rotmat'].reshape(batch_size, seqlen, -1, 3, 3)
        return output
