"""Weak-perspective camera projection (reference: lib/models/spin.py:113-157).  ATen version for the
training graph; inference uses maed_smpl_joints_project_fwd."""
import torch


def projection(pred_joints, pred_camera):
    t = torch.stack([pred_camera[:, 1], pred_camera[:, 2], 2 * 5000. / (224. * pred_camera[:, 0] + 1e-9)], dim=-1)
    pts = pred_joints + t.unsqueeze(1)
    proj = pts / pts[:, :, -1].unsqueeze(-1)
    return (5000. * proj[:, :, :2]) / (224. / 2.)
