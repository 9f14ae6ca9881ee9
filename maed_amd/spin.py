"""lib/models/spin.py: weak-perspective camera projection (:113-157; ATen version for CPU tensors and a
differentiable J_regressor override -- the GPU paths use maed_smpl_joints_project_fwd / tail.SmplTailFn) and the
iterative `Regressor` (:17-110), which lives in maed_amd/iterative.py and is re-exported here lazily because it
builds on KTD, which imports this module."""
import torch


def __getattr__(name):
    if name == "Regressor":
        from .iterative import Regressor
        return Regressor
    raise AttributeError(name)


def projection(pred_joints, pred_camera):
    t = torch.stack([pred_camera[:, 1], pred_camera[:, 2], 2 * 5000. / (224. * pred_camera[:, 0] + 1e-9)], dim=-1)
    pts = pred_joints + t.unsqueeze(1)
    proj = pts / pts[:, :, -1].unsqueeze(-1)
    return (5000. * proj[:, :, :2]) / (224. / 2.)
