"""The train-step and checkpoint semantics of the reference's driver (lib/core/trainer.py) for the MI355X modules --
SURVEY 8(f) rank 2, host logic only: what `Trainer.train` does between fetching the batches and logging, and what
`save_model` / `resume_pretrained` write and read.  The reference's own `Trainer` also runs unchanged on top of
`maed_amd.MAED` + `maed_amd.loss.Loss` + `maed_amd.ddp.FusedAdam`; this module is the same logic without
its per-iteration host synchronisations (`.item()` on every loss term, trainer.py:209-211,270-276).

    step = TrainStep(model, criterion, optimizer)
    total, terms = step(target_2d=..., target_3d=..., target_img=...)       # trainer.py:159-204, 240-262

Batches are the reference's dicts: target_2d {'images' (N2,T,3,H,W), 'kp_2d'}, target_3d {'images', 'kp_2d', 'kp_3d',
'theta', 'w_smpl'}, target_img {'image' (N,3,H,W), 'kp_2d', 'theta', 'w_smpl'[, 'kp_3d']}.
"""
import torch


class TrainStep:
    def __init__(self, model, criterion, optimizer):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer

    def losses(self, target_2d=None, target_3d=None, target_img=None):
        """two forwards, frame-count weighted sum (trainer.py:159-204) -> (loss to back-propagate, merged total, merged terms)"""
        loss_vid, dict_vid, nt_vid = 0, {}, 0
        if target_2d or target_3d:
            if target_2d and target_3d:
                inp_vid = torch.cat((target_2d['images'], target_3d['images']), dim=0)     # 2D-only clips in front (:159-160)
            else:
                inp_vid = (target_3d or target_2d)['images']
            nt_vid = inp_vid.shape[0] * inp_vid.shape[1]
            loss_vid, dict_vid = self.criterion(preds=self.model(inp_vid), target_3d=target_3d, target_2d=target_2d)
        loss_img, dict_img, nt_img = 0, {}, 0
        if target_img:
            inp_img = target_img['image'].unsqueeze(1)                                       # (:170-172) T = 1
            nt_img = inp_img.shape[0]
            loss_img, dict_img = self.criterion(preds=self.model(inp_img), target_img=target_img)
        w_vid = nt_vid / max(nt_img + nt_vid, 1)
        w_img = 1 - w_vid
        total, terms = self.criterion.merge_loss(loss_vid, dict_vid, loss_img, dict_img, vid_w=w_vid, img_w=w_img)
        return loss_img * w_img + loss_vid * w_vid, total, terms

    def __call__(self, target_2d=None, target_3d=None, target_img=None):
        loss, total, terms = self.losses(target_2d, target_3d, target_img)
        self.optimizer.zero_grad()                                                           # trainer.py:240-248
        loss.backward()
        self.optimizer.step()
        return total, terms


def save_checkpoint(path, model, optimizer, epoch, performance, ddp_prefix=True):
    """trainer.py:330-352: {'epoch', 'state_dict', 'performance', 'optimizer'}; the reference saves the DDP-wrapped model, so
    its keys carry a 'module.' prefix (eval.py:29 strips it) -- kept by default for interchangeability"""
    sd = model.state_dict()
    if ddp_prefix:
        sd = {'module.' + k: v for k, v in sd.items()}
    torch.save({'epoch': epoch, 'state_dict': sd, 'performance': performance, 'optimizer': optimizer.state_dict()}, path)


def load_checkpoint(path, model, optimizer=None, strict=True):
    """trainer.py:354-368 (resume) / eval.py:24-31 (evaluation: optimizer=None).  Returns (epoch, performance).
    `decoder.smpl.*` buffers are skipped as the reference does (eval.py:29, train.py:101): they come from the SMPL model
    file, not from training."""
    ckpt = torch.load(path, map_location='cpu')
    sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in ckpt['state_dict'].items()}
    sd = {k: v for k, v in sd.items() if not k.startswith('decoder.smpl')}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [k for k in missing if not k.startswith('decoder.smpl')]
    if strict and (missing or unexpected):
        raise RuntimeError(f"checkpoint mismatch: missing {missing[:5]}..., unexpected {list(unexpected)[:5]}...")
    if optimizer is not None and 'optimizer' in ckpt:
        optimizer.load_state_dict(ckpt['optimizer'])
    from . import ops
    ops.bump_weight_epoch()
    return ckpt.get('epoch', 0), ckpt.get('performance')
