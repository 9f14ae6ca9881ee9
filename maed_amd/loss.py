"""Training objective (reference: lib/core/loss.py) -- the caller right after MAED.forward in
lib/core/trainer.py:253-262.  Same classes, constructor arguments, call signatures and return values:

    Loss(e_loss_weight=60., e_3d_loss_weight=30., e_pose_loss_weight=1., e_shape_loss_weight=0.001,
         e_smpl_norm_loss=1., e_smpl_accl_loss=0., device='cuda')
    Loss.forward(preds, target_3d=..., target_2d=...) | (preds, target_img=...) -> (loss, loss_dict)
    Loss.merge_loss(loss_vid, loss_vid_dict, loss_img, loss_img_dict, vid_w, img_w)

On HIP tensors LossVideo / LossImage run as ONE fused forward+backward (maed_loss_fwd_bwd: two launches that
produce the five weighted terms, their sum and d total / d preds) instead of ~90 ATen launches; the entries of
loss_dict are then detached views of one device vector (they are only logged: trainer.py:209-211); the optional acceleration
term (e_smpl_accl_loss > 0, off in every shipped config) is one more launch (maed_loss_accl_fwd_bwd).  The ATen composition
below is the same arithmetic for HOST tensors only -- a device tensor never reaches it.
"""
import torch
import torch.nn as nn

from . import tail


def batch_rodrigues(axisang):
    """lib/utils/geometry.py:12-56 (axis-angle (N,3) -> rotation matrices, flattened (N,9))"""
    angle = torch.norm(axisang + 1e-8, p=2, dim=1, keepdim=True)
    normalized = axisang / angle
    half = angle * 0.5
    quat = torch.cat([torch.cos(half), torch.sin(half) * normalized], dim=1)
    quat = quat / quat.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = quat[:, 0], quat[:, 1], quat[:, 2], quat[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1)


def _on_dev(t):
    from . import ops
    return ops.on_library_device(t)


class _LossBase(nn.Module):
    def __init__(self, device='cuda'):
        super().__init__()
        self.device = device

    def _zero(self, like=None):
        return torch.zeros((), dtype=torch.float32, device=like.device if like is not None else self.device)

    def keypoint_loss(self, pred_keypoints_2d, gt_keypoints_2d):
        """loss.py:21-38: confidence-weighted squared error, mean over every coordinate"""
        if len(gt_keypoints_2d) == 0:
            return self._zero(pred_keypoints_2d)
        if gt_keypoints_2d.dim() > 3:
            gt_keypoints_2d = gt_keypoints_2d.reshape((-1,) + gt_keypoints_2d.shape[2:])
            pred_keypoints_2d = pred_keypoints_2d.reshape((-1,) + pred_keypoints_2d.shape[2:])
        conf = gt_keypoints_2d[:, :, -1:]
        return (conf * (pred_keypoints_2d - gt_keypoints_2d[:, :, :-1]) ** 2).mean()

    def keypoint_3d_loss(self, pred_keypoints_3d, gt_keypoints_3d):
        """loss.py:40-62: both sides centred on the pelvis = midpoint of joints 27 and 28"""
        if len(gt_keypoints_3d) == 0:
            return self._zero(pred_keypoints_3d)
        if gt_keypoints_3d.dim() > 3:
            gt_keypoints_3d = gt_keypoints_3d.reshape((-1,) + gt_keypoints_3d.shape[2:])
            pred_keypoints_3d = pred_keypoints_3d.reshape((-1,) + pred_keypoints_3d.shape[2:])
        conf = gt_keypoints_3d[:, :, -1:]
        gt = gt_keypoints_3d[:, :, :-1]
        gt = gt - ((gt[:, 27] + gt[:, 28]) / 2)[:, None]
        pred = pred_keypoints_3d - ((pred_keypoints_3d[:, 27] + pred_keypoints_3d[:, 28]) / 2)[:, None]
        return (conf * (pred - gt) ** 2).mean()

    def smpl_losses(self, pred_pose, pred_shape, gt_pose, gt_shape, w_smpl):
        """loss.py:64-92: MSE on rotation matrices and betas.  Video input (N,T,.) keeps only the frames whose SMPL labels are
        valid (:77-82); image input (N,.) is NOT masked by the reference (w_smpl is ignored there) and neither is it here."""
        if pred_pose.dim() > 2:
            w_smpl = w_smpl.reshape(-1)
            pred_pose, pred_shape = pred_pose.reshape(-1, pred_pose.shape[-1])[w_smpl], pred_shape.reshape(-1, pred_shape.shape[-1])[w_smpl]
            gt_pose, gt_shape = gt_pose.reshape(-1, gt_pose.shape[-1])[w_smpl], gt_shape.reshape(-1, gt_shape.shape[-1])[w_smpl]
        if len(pred_pose) == 0:
            return self._zero(pred_pose), self._zero(pred_pose)
        pred_rot = batch_rodrigues(pred_pose.reshape(-1, 3)).reshape(-1, 24, 3, 3)
        gt_rot = batch_rodrigues(gt_pose.reshape(-1, 3)).reshape(-1, 24, 3, 3)
        return ((pred_rot - gt_rot) ** 2).mean(), ((pred_shape - gt_shape) ** 2).mean()

    def accl_losses(self, pred_keypoints_3d, gt_keypoints_3d):
        """loss.py:94-117: second temporal difference of the joints, weighted by the (squared, twice) confidences"""
        if len(pred_keypoints_3d) == 0:
            return self._zero(pred_keypoints_3d)
        conf = gt_keypoints_3d[:, :, :, -1:]
        conf_velocity = conf[:, 1:] * conf[:, 1:]
        conf_accl = conf_velocity[:, 1:] * conf_velocity[:, 1:]
        pv = pred_keypoints_3d[:, 1:] - pred_keypoints_3d[:, :-1]
        pa = (pv[:, 1:] - pv[:, :-1]) * conf_accl
        g = gt_keypoints_3d[:, :, :, :3]
        gv = g[:, 1:] - g[:, :-1]
        ga = (gv[:, 1:] - gv[:, :-1]) * conf_accl
        return ((pa - ga) ** 2).mean()

    # ---- shared by LossVideo / LossImage ---------------------------------------------------------------------------
    def _terms_aten(self, pred_j2d, gt_j2d, pred_j3d, gt_j3d, pred_theta, gt_theta, w_smpl, w3d, want_accl=0.0, accl_args=None):
        d = {'loss_kp_2d': self.e_loss_weight * self.keypoint_loss(pred_j2d, gt_j2d),
             'loss_kp_3d': (w3d * self.keypoint_3d_loss(pred_j3d, gt_j3d)) if gt_j3d is not None else self._zero(pred_j2d)}
        if self.e_shape_loss_weight > 0 and self.e_pose_loss_weight > 0:
            lp, ls = self.smpl_losses(pred_theta[..., 3:75], pred_theta[..., 75:], gt_theta[..., 3:75], gt_theta[..., 75:], w_smpl)
            d['loss_shape'] = ls * self.e_shape_loss_weight
            d['loss_pose'] = lp * self.e_pose_loss_weight
        if self.e_smpl_norm_loss > 0:
            flat = pred_theta.reshape(-1, pred_theta.shape[-1])
            d['loss_norm'] = self.e_smpl_norm_loss * torch.norm(flat[:, 3:], p=2, dim=(0, 1)) / flat.shape[0]
        if want_accl > 0:
            d['loss_accl'] = want_accl * self.accl_losses(*accl_args)
        return torch.stack(list(d.values())).sum(), d

    def _terms_fused(self, pred_j2d, gt_j2d, pred_j3d_all, gt_j3d, pred_theta_all, gt_theta, w_smpl, skip_frames, w3d):
        both = self.e_shape_loss_weight > 0 and self.e_pose_loss_weight > 0
        weights = (self.e_loss_weight, w3d if gt_j3d is not None else 0.0, self.e_pose_loss_weight if both else 0.0,
                   self.e_shape_loss_weight if both else 0.0, max(self.e_smpl_norm_loss, 0.0))
        flat = lambda t: None if t is None else t.reshape((-1,) + t.shape[-2:]) if t.dim() > 3 else t
        flat_th = lambda t: t.reshape(-1, t.shape[-1])
        total, losses = tail.FusedLossFn.apply(flat(pred_j2d), flat(pred_j3d_all), flat_th(pred_theta_all), flat(gt_j2d), flat(gt_j3d),
                                               flat_th(gt_theta), w_smpl.reshape(-1), skip_frames, weights)
        d = {'loss_kp_2d': losses[0], 'loss_kp_3d': losses[1]}
        if both:
            d['loss_shape'], d['loss_pose'] = losses[2], losses[3]
        if self.e_smpl_norm_loss > 0:
            d['loss_norm'] = losses[4]
        return total, d


class LossVideo(_LossBase):
    """loss.py:121-211.  preds: kp_2d (N,T,49,2), kp_3d (N,T,49,3), theta (N,T,85); data_3d: kp_2d (N3,T,49,3),
    kp_3d (N3,T,49,4), theta (N3,T,85), w_smpl (N3,T); data_2d (optional): kp_2d (N2,T,49,3) -- the first N2 clips of
    preds are the 2D-only ones."""

    def __init__(self, e_loss_weight=60., e_3d_loss_weight=30., e_pose_loss_weight=1., e_shape_loss_weight=0.001,
                 e_smpl_norm_loss=1., e_smpl_accl_loss=0., device='cuda'):
        super().__init__(device)
        self.e_loss_weight, self.e_3d_loss_weight = e_loss_weight, e_3d_loss_weight
        self.e_pose_loss_weight, self.e_shape_loss_weight = e_pose_loss_weight, e_shape_loss_weight
        self.e_smpl_norm_loss, self.e_smpl_accl_loss = e_smpl_norm_loss, e_smpl_accl_loss
        self.e_3d_loss_weight_branch2 = 300.
        self.e_loss_weight_branch2 = 300.

    def forward(self, preds, data_3d, data_2d):
        if data_2d:
            n2 = data_2d['kp_2d'].shape[0]
            gt_j2d = torch.cat((data_2d['kp_2d'], data_3d['kp_2d']), 0)
        else:
            n2 = 0
            gt_j2d = data_3d['kp_2d']
        w_smpl = data_3d['w_smpl'].type(torch.bool)
        if _on_dev(preds['kp_2d']):
            T = preds['kp_3d'].shape[1]
            total, d = self._terms_fused(preds['kp_2d'], gt_j2d, preds['kp_3d'], data_3d['kp_3d'], preds['theta'], data_3d['theta'], w_smpl,
                                         n2 * T, self.e_3d_loss_weight)
            if self.e_smpl_accl_loss > 0:      # the acceleration term as its own launch (maed_loss_accl_fwd_bwd); needs clips of >= 3 frames
                d['loss_accl'] = tail.AcclLossFn.apply(preds['kp_3d'][n2:], data_3d['kp_3d'], self.e_smpl_accl_loss)
                total = total + d['loss_accl']
            return total, d
        pred_j3d, pred_theta = preds['kp_3d'][n2:], preds['theta'][n2:]
        return self._terms_aten(preds['kp_2d'], gt_j2d, pred_j3d, data_3d['kp_3d'], pred_theta, data_3d['theta'], w_smpl,
                                self.e_3d_loss_weight, self.e_smpl_accl_loss, (pred_j3d, data_3d['kp_3d']))


class LossImage(_LossBase):
    """loss.py:215-283.  preds carry a singleton T axis (squeezed); target: kp_2d (N,49,3), theta (N,85), w_smpl (N,),
    optional kp_3d (N,49,4)."""

    def __init__(self, e_loss_weight=60., e_3d_loss_weight=600., e_pose_loss_weight=1., e_shape_loss_weight=0.001,
                 e_smpl_norm_loss=1., device='cuda'):
        super().__init__(device)
        self.e_loss_weight, self.e_3d_loss_weight = e_loss_weight, e_3d_loss_weight
        self.e_pose_loss_weight, self.e_shape_loss_weight = e_pose_loss_weight, e_shape_loss_weight
        self.e_smpl_norm_loss = e_smpl_norm_loss
        self.e_loss_weight_branch2 = 300.

    def forward(self, preds, target):
        gt_j3d = target['kp_3d'] if 'kp_3d' in target else None
        pred_j2d, pred_j3d, pred_theta = preds['kp_2d'].squeeze(1), preds['kp_3d'].squeeze(1), preds['theta'].squeeze(1)
        w_smpl = target['w_smpl'].type(torch.bool)
        if _on_dev(pred_j2d):
            # the reference's smpl_losses masks by w_smpl for video input only (loss.py:77): every image counts here
            return self._terms_fused(pred_j2d, target['kp_2d'], pred_j3d, gt_j3d, pred_theta, target['theta'], torch.ones_like(w_smpl), 0,
                                     self.e_3d_loss_weight)
        return self._terms_aten(pred_j2d, target['kp_2d'], pred_j3d, gt_j3d, pred_theta, target['theta'], w_smpl, self.e_3d_loss_weight)


class Loss(nn.Module):
    """loss.py:285-345"""

    def __init__(self, e_loss_weight=60., e_3d_loss_weight=30., e_pose_loss_weight=1., e_shape_loss_weight=0.001,
                 e_smpl_norm_loss=1., e_smpl_accl_loss=0., device='cuda'):
        super().__init__()
        self.loss_video = LossVideo(e_loss_weight, e_3d_loss_weight, e_pose_loss_weight, e_shape_loss_weight, e_smpl_norm_loss,
                                    e_smpl_accl_loss, device)
        self.loss_image = LossImage(e_loss_weight, e_3d_loss_weight, e_pose_loss_weight, e_shape_loss_weight, e_smpl_norm_loss, device)

    def forward(self, preds, **kwargs):
        if 'target_2d' in kwargs:
            return self.loss_video(preds, kwargs['target_3d'], kwargs['target_2d'])
        if 'target_img' in kwargs:
            return self.loss_image(preds, kwargs['target_img'])
        return 0, {}

    def merge_loss(self, loss_vid, loss_vid_dict, loss_img, loss_img_dict, vid_w=1.0, img_w=1.0):
        loss_dict = {}
        for k in set(list(loss_vid_dict.keys()) + list(loss_img_dict.keys())):
            v = 0
            if k in loss_vid_dict:
                v = v + loss_vid_dict[k] * vid_w
            if k in loss_img_dict:
                v = v + loss_img_dict[k] * img_w
            loss_dict[k] = v
        return loss_vid * vid_w + loss_img * img_w, loss_dict
