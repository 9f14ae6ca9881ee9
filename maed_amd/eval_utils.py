"""Evaluation metrics (reference: lib/utils/eval_utils.py), same function names, torch tensors in and out, computed by
libmaed_hip.so (maed_amd/csrc/eval_metrics.hip) on the device the tensors live on -- the reference does this part in
numpy / torch on the CPU after copying every prediction back.

    compute_accel(joints)                                   eval_utils.py:10-21
    compute_error_accel(joints_gt, joints_pred, vis=None)   eval_utils.py:24-52
    compute_error_verts(pred_verts, target_verts=None, target_theta=None, smpl=None)   eval_utils.py:55-90
    batch_compute_similarity_transform_torch(S1, S2)        eval_utils.py:201-252
    pose_errors(pred_j3d, target_j3d)                       lib/core/evaluate.py:139-160 in one launch
"""
import torch

from . import _lib as L
from . import ops


def _dev(t, name):
    """fp32 contiguous tensor the library can address (a GPU tensor; under tests/hostsim a host tensor)"""
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    if not (t.is_cuda or L.lib().maed_version() < 0):
        raise RuntimeError(f"{name}: evaluation metrics run on the GPU (libmaed_hip); move the tensor to a cuda device")
    return t.float().contiguous()


def compute_accel(joints):
    """(N,J,3) -> (N-2,) mean joint acceleration"""
    joints = _dev(joints, "compute_accel")
    N, J = joints.shape[:2]
    out = torch.empty(max(N - 2, 0), dtype=torch.float32, device=joints.device)
    if N < 3:
        return out
    ops.check(L.lib().maed_eval_accel(ops._p(joints), None, N, J, ops._p(out), ops._stream()), "eval_accel")
    return out


def compute_error_accel(joints_gt, joints_pred, vis=None):
    """(N,J,3) x2 -> (N-2,) acceleration error; `vis` (N,) drops every window touching an invisible frame (eval_utils.py:43-50)"""
    joints_gt, joints_pred = _dev(joints_gt, "compute_error_accel"), _dev(joints_pred, "compute_error_accel")
    N, J = joints_pred.shape[:2]
    out = torch.empty(max(N - 2, 0), dtype=torch.float32, device=joints_pred.device)
    if N < 3:
        return out
    ops.check(L.lib().maed_eval_accel(ops._p(joints_pred), ops._p(joints_gt), N, J, ops._p(out), ops._stream()), "eval_accel")
    if vis is not None:
        invis = ~torch.as_tensor(vis, dtype=torch.bool, device=out.device)
        bad = invis | torch.roll(invis, -1) | torch.roll(invis, -2)
        out = out[~bad[:-2]]
    return out


def target_vertices(target_theta, smpl, chunk=5000):
    """SMPL vertices of (N,85) theta = [cam(3) | axis-angle pose(72) | betas(10)] (eval_utils.py:66-84, pose2rot=True)"""
    from .loss import batch_rodrigues
    theta = _dev(target_theta, "target_vertices")
    out = []
    for th in torch.split(theta, chunk):
        rot = batch_rodrigues(th[:, 3:75].reshape(-1, 3)).reshape(-1, 24, 3, 3)
        out.append(smpl.lbs_hip(th[:, 75:].contiguous(), rot)[0])
    return torch.cat(out, dim=0)


def compute_error_verts(pred_verts, target_verts=None, target_theta=None, smpl=None):
    """(N,6890,3) -> (N,) mean per-vertex error; target vertices from `target_theta` through `smpl` when not given"""
    pred_verts = _dev(pred_verts, "compute_error_verts")
    if target_verts is None:
        if smpl is None:
            raise ValueError("compute_error_verts: pass target_verts, or target_theta together with the SMPL module")
        target_verts = target_vertices(target_theta, smpl)
    target_verts = _dev(target_verts, "compute_error_verts")
    assert len(pred_verts) == len(target_verts)
    N, V = pred_verts.shape[:2]
    out = torch.empty(N, dtype=torch.float32, device=pred_verts.device)
    ops.check(L.lib().maed_eval_vertex_error(ops._p(pred_verts), ops._p(target_verts), N, V, ops._p(out), ops._stream()), "eval_vertex_error")
    return out


def batch_compute_similarity_transform_torch(S1, S2):
    """S1_hat = s R S1 + t closest to S2 (orthogonal Procrustes with scale, det R = +1); (N,J,3) or (N,3,J) like the reference"""
    if S1.shape[-1] != 3 and S1.shape[1] != 3:
        raise NotImplementedError("only 3-D point sets are evaluated on the MAED path")
    transposed = S1.shape[-1] != 3          # (N,3,J) input; evaluate.py:159 passes (N,J,3)
    if transposed:
        S1, S2 = S1.permute(0, 2, 1), S2.permute(0, 2, 1)
    S1, S2 = _dev(S1, "similarity_transform"), _dev(S2, "similarity_transform")
    N, J = S1.shape[:2]
    out = torch.empty_like(S1)
    ops.check(L.lib().maed_similarity_transform(ops._p(S1), ops._p(S2), N, J, ops._p(out), ops._stream()), "similarity_transform")
    return out.permute(0, 2, 1) if transposed else out


def pose_errors(pred_j3d, target_j3d):
    """evaluate.py:139-160 for (N,J,3) predictions and (N,J,4) targets [x,y,z,vis]:
    -> mpjpe (N,), pa_mpjpe (N,), and the masked pelvis-centred joints pred_c, target_c (N,J,3)"""
    pred, tgt = _dev(pred_j3d, "pose_errors"), _dev(target_j3d, "pose_errors")
    N, J = pred.shape[:2]
    assert tgt.shape == (N, J, 4), tgt.shape
    dev = pred.device
    mpjpe, pa = torch.empty(N, dtype=torch.float32, device=dev), torch.empty(N, dtype=torch.float32, device=dev)
    pred_c, tgt_c = torch.empty(N, J, 3, dtype=torch.float32, device=dev), torch.empty(N, J, 3, dtype=torch.float32, device=dev)
    ops.check(L.lib().maed_eval_pose_errors(ops._p(pred), ops._p(tgt), N, J, ops._p(mpjpe), ops._p(pa), ops._p(pred_c), ops._p(tgt_c),
                                            ops._stream()), "eval_pose_errors")
    return mpjpe, pa, pred_c, tgt_c
