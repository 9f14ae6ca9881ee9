"""6D rotation / axis-angle helpers (reference: lib/utils/geometry.py:320-334,58-87,143-223,90-140) as
differentiable ATen compositions for HOST tensors: what the `-m "not gpu"` suite compares the kernels with.  On a library device both the inference
and the training path run maed_rot6d_pose_fwd / _bwd (tail.SmplTailFn)."""
import torch
import torch.nn.functional as F


def rot6d_to_rotmat(x):
    x = x.reshape(-1, 3, 2)
    b1 = F.normalize(x[:, :, 0], dim=1, eps=1e-6)
    dot = torch.sum(b1 * x[:, :, 1], dim=1, keepdim=True)
    b2 = F.normalize(x[:, :, 1] - dot * b1, dim=-1, eps=1e-6)
    b3 = torch.cross(b1, b2, dim=1)
    return torch.stack([b1, b2, b3], dim=-1)


def rotation_matrix_to_quaternion(R, eps=1e-6):
    r = R.transpose(1, 2)
    d2 = r[:, 2, 2] < eps
    d0d1 = r[:, 0, 0] > r[:, 1, 1]
    d0nd1 = r[:, 0, 0] < -r[:, 1, 1]
    t0 = 1 + r[:, 0, 0] - r[:, 1, 1] - r[:, 2, 2]
    q0 = torch.stack([r[:, 1, 2] - r[:, 2, 1], t0, r[:, 0, 1] + r[:, 1, 0], r[:, 2, 0] + r[:, 0, 2]], -1)
    t1 = 1 - r[:, 0, 0] + r[:, 1, 1] - r[:, 2, 2]
    q1 = torch.stack([r[:, 2, 0] - r[:, 0, 2], r[:, 0, 1] + r[:, 1, 0], t1, r[:, 1, 2] + r[:, 2, 1]], -1)
    t2 = 1 - r[:, 0, 0] - r[:, 1, 1] + r[:, 2, 2]
    q2 = torch.stack([r[:, 0, 1] - r[:, 1, 0], r[:, 2, 0] + r[:, 0, 2], r[:, 1, 2] + r[:, 2, 1], t2], -1)
    t3 = 1 + r[:, 0, 0] + r[:, 1, 1] + r[:, 2, 2]
    q3 = torch.stack([t3, r[:, 1, 2] - r[:, 2, 1], r[:, 2, 0] - r[:, 0, 2], r[:, 0, 1] - r[:, 1, 0]], -1)
    c0 = (d2 & d0d1)[:, None].to(R.dtype)
    c1 = (d2 & ~d0d1)[:, None].to(R.dtype)
    c2 = (~d2 & d0nd1)[:, None].to(R.dtype)
    c3 = (~d2 & ~d0nd1)[:, None].to(R.dtype)
    q = q0 * c0 + q1 * c1 + q2 * c2 + q3 * c3
    q = q / torch.sqrt(t0[:, None] * c0 + t1[:, None] * c1 + t2[:, None] * c2 + t3[:, None] * c3)
    return q * 0.5


def quaternion_to_angle_axis(q):
    q1, q2, q3 = q[..., 1], q[..., 2], q[..., 3]
    sin_sq = q1 * q1 + q2 * q2 + q3 * q3
    sin_t = torch.sqrt(sin_sq)
    cos_t = q[..., 0]
    two_theta = 2.0 * torch.where(cos_t < 0.0, torch.atan2(-sin_t, -cos_t), torch.atan2(sin_t, cos_t))
    k = torch.where(sin_sq > 0.0, two_theta / sin_t, 2.0 * torch.ones_like(sin_t))
    return torch.stack([q1 * k, q2 * k, q3 * k], dim=-1)


def rotation_matrix_to_angle_axis(R):
    aa = quaternion_to_angle_axis(rotation_matrix_to_quaternion(R.reshape(-1, 3, 3)))
    return torch.where(torch.isnan(aa), torch.zeros_like(aa), aa)
