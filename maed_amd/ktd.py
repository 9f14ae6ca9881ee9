"""Kinematic Topology Decoder (reference: lib/models/ktd.py).

Same module/parameter names (`fc1, drop1, fc2, drop2, joint_regs.{0..23}, decshape, deccam, smpl`).
Inference (eval mode, no grad) runs on libmaed_hip.so: two f32 GEMMs, ONE GEMM for the 1024-wide
feature part of all 24 joint regressors + shape + cam, the serial ancestor chain in one kernel
(maed_ktd_chain_fwd, replaces 24 cat+Linear launches, ktd.py:81-84), fused 6D->rotmat->axis-angle,
SMPL LBS, the joint-regressor GEMM on f32 MFMA, the int64 joint_map gather and the projection.
Training on the GPU runs the same forward kernels inside two autograd Functions (maed_amd/tail.py) whose backward
is ~10 hand-written launches; fc1/fc2 + Dropout stay ATen (two plain GEMMs and the framework's RNG).  The
ATen composition of the whole tail is kept for CPU tensors and for a differentiable J_regressor override.
"""
import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .geometry import rot6d_to_rotmat, rotation_matrix_to_angle_axis
from .smpl import SMPL
from .spin import projection
from . import tail

# lib/models/ktd.py:10-35
ANCESTOR_INDEX = [
    [], [0], [0], [0], [0, 1], [0, 2], [0, 3], [0, 1, 4], [0, 2, 5], [0, 3, 6],
    [0, 1, 4, 7], [0, 2, 5, 8], [0, 3, 6, 9], [0, 3, 6, 9], [0, 3, 6, 9],
    [0, 3, 6, 9, 12], [0, 3, 6, 9, 13], [0, 3, 6, 9, 14],
    [0, 3, 6, 9, 13, 16], [0, 3, 6, 9, 14, 17],
    [0, 3, 6, 9, 13, 16, 18], [0, 3, 6, 9, 14, 17, 19],
    [0, 3, 6, 9, 13, 16, 18, 20], [0, 3, 6, 9, 14, 17, 19, 21],
]


class KTD(nn.Module):
    def __init__(self, feat_dim=2048, hidden_dim=1024, smpl_arrays=None, **kwargs):
        super().__init__()
        self.feat_dim, self.hidden_dim = feat_dim, hidden_dim
        self.smpl = SMPL(smpl_arrays)
        self.fc1 = nn.Linear(feat_dim, hidden_dim)
        self.drop1 = nn.Dropout()
        self.fc2 = nn.Linear(hidden_dim, hidden_dim)
        self.drop2 = nn.Dropout()
        self.joint_regs = nn.ModuleList()
        for anc in ANCESTOR_INDEX:
            reg = nn.Linear(hidden_dim + 6 * len(anc), 6)
            nn.init.xavier_uniform_(reg.weight, gain=0.01)
            self.joint_regs.append(reg)
        self.decshape = nn.Linear(hidden_dim, 10)
        self.deccam = nn.Linear(hidden_dim, 3)
        nn.init.xavier_uniform_(self.decshape.weight, gain=0.01)
        nn.init.xavier_uniform_(self.deccam.weight, gain=0.01)
        self._packed_key, self._packed = None, None
        self._fc_cache, self._fc_cache2 = ops.WeightCache(), ops.WeightCache()      # fp32 [out,in] master + transposed image of fc1 / fc2
        # engine of the head's fp32 matrix products (fc1, fc2, the packed regressors and their gradients): None = the process-wide fp32 mode (exact VALU kernels by
        # default); "bf16x3" = split-bf16 products on the matrix cores (error ~2^-16 of |a||b|: still 2^8 below a bf16 product) -- what MAED sets for a bf16-mode
        # model, whose encoder output is bf16-accurate anyway; the fp32 model keeps the exact kernels
        self.head_matmul = None
        self._pending_backwards = 0
        self.grads_ready = None  # callback(self) set by the data-parallel gradient bucketer

    def _regressors(self):
        return list(self.joint_regs) + [self.decshape, self.deccam]

    def fused_parameters(self):
        """parameters whose gradients tail.KtdChainFn writes straight into .grad (maed_ktd_unpack_add)"""
        return [t for m in self._regressors() for t in (m.weight, m.bias)]

    def _ptr_table(self, grads):
        t = L.KtdPtrs()
        for j, m in enumerate(self._regressors()):
            t.w[j], t.b[j] = ops._p(m.weight), ops._p(m.bias)
            if grads:
                for p in (m.weight, m.bias):
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                t.gw[j], t.gb[j] = ops._p(m.weight.grad), ops._p(m.bias.grad)
        return t

    def _head_train(self, x):
        """ktd.py:71-86 with the 26 small regressors as ONE packed GEMM + the chain kernel (differentiable)"""
        from . import ste_modes
        x = x.float()
        hm = self.head_matmul
        x = ste_modes.dropout(ste_modes.LinearTokFn.apply(x, self.fc1.weight, self.fc1.bias, self._fc_cache, True, hm, True), self.drop1.p, self.drop1.training)
        x = ste_modes.dropout(ste_modes.LinearTokFn.apply(x, self.fc2.weight, self.fc2.bias, self._fc_cache2, True, hm, True), self.drop2.p, self.drop2.training)
        return tail.KtdChainFn.apply(x, self, *self.fused_parameters())

    # ---- ATen composition for host tensors (the CPU suite's comparison arm; a library device takes _head_hip / _head_train) ----
    def _head_torch(self, x):
        x = self.drop1(self.fc1(x))
        x = self.drop2(self.fc2(x))
        shape, cam = self.decshape(x), self.deccam(x)
        pose = []
        for anc, reg in zip(ANCESTOR_INDEX, self.joint_regs):
            pose.append(reg(torch.cat([x] + [pose[i] for i in anc], dim=1)))
        return torch.cat(pose, dim=1), shape, cam

    # ---- HIP path (inference) ---------------------------------------------------------------------
    def _pack(self):
        key = (ops.WEIGHT_EPOCH, tuple(p._version for p in self.parameters()), self.fc1.weight.data_ptr())
        if key != self._packed_key:
            h = self.hidden_dim
            with torch.no_grad():
                w_feat = torch.cat([r.weight[:, :h] for r in self.joint_regs] + [self.decshape.weight, self.deccam.weight], 0).contiguous()
                b_feat = torch.cat([r.bias for r in self.joint_regs] + [self.decshape.bias, self.deccam.bias], 0).contiguous()
                w_anc = torch.cat([r.weight[:, h:].reshape(-1) for r in self.joint_regs[1:]], 0).contiguous()
            self._packed_key, self._packed = key, (w_feat, b_feat, w_anc)
        return self._packed

    def _head_hip(self, x):
        w_feat, b_feat, w_anc = self._pack()
        x = x.float().contiguous()
        hm = self.head_matmul
        h1 = ops.gemm_nt(x, self.fc1.weight.detach(), L.EPI_STORE, bias=self.fc1.bias, prec=hm)
        h2 = ops.gemm_nt(h1, self.fc2.weight.detach(), L.EPI_STORE, bias=self.fc2.bias, prec=hm)
        out = ops.gemm_nt(h2, w_feat, L.EPI_STORE, bias=b_feat, prec=hm)             # (F, 144 + 10 + 3)
        base = out[:, :144].contiguous()
        pose = torch.empty_like(base)
        ops.check(L.lib().maed_ktd_chain_fwd(ops._p(base), ops._p(w_anc), ops._p(pose), x.shape[0], ops._stream()), "ktd_chain_fwd")
        return pose, out[:, 144:154].contiguous(), out[:, 154:157].contiguous()

    def _use_hip(self, x):
        return ops.on_library_device(x) and not self.training and not (torch.is_grad_enabled() and (x.requires_grad or self.fc1.weight.requires_grad))

    def _use_hip_train(self, x, J_regressor):
        return ops.on_library_device(x) and J_regressor is None and not self._use_hip(x)

    def forward(self, x, seqlen, J_regressor=None, return_shape_cam=False, **kwargs):
        hip = self._use_hip(x)
        hip_train = self._use_hip_train(x, J_regressor)
        if ops.on_library_device(x) and not hip and not hip_train:
            # differentiable graph + an evaluation joint regressor: the reference only ever passes J_regressor under no_grad
            # (lib/core/evaluate.py:78); there is no kernel for the regressor's backward and no silent ATen detour on the device
            raise NotImplementedError("KTD: J_regressor is an inference-time input (call under torch.no_grad(), as lib/core/evaluate.py:78 does)")
        if hip:
            pred_pose, pred_shape, pred_cam = self._head_hip(x)
        elif hip_train:
            pred_pose, pred_shape, pred_cam = self._head_train(x.float())
        else:
            pred_pose, pred_shape, pred_cam = self._head_torch(x.float())
        if return_shape_cam:
            return pred_shape, pred_cam
        if hip_train:
            theta, verts, kp2d, kp3d, rotmat = tail.SmplTailFn.apply(pred_pose, pred_shape, pred_cam, self.smpl)
            return dict(theta=theta, verts=verts, kp_2d=kp2d, kp_3d=kp3d, rotmat=rotmat)
        return self.get_output(pred_pose, pred_shape, pred_cam, J_regressor, hip)

    def get_output(self, pred_pose, pred_shape, pred_cam, J_regressor, hip=None):
        """ktd.py:94-124"""
        nt = pred_pose.shape[0]
        hip = self._use_hip(pred_pose) if hip is None else hip
        if hip:
            rotmat = torch.empty(nt, 24, 3, 3, dtype=torch.float32, device=pred_pose.device)
            aa = torch.empty(nt, 72, dtype=torch.float32, device=pred_pose.device)
            ops.check(L.lib().maed_rot6d_pose_fwd(ops._p(pred_pose.contiguous()), ops._p(rotmat), ops._p(aa), nt * 24, ops._stream()), "rot6d_pose_fwd")
            verts, j24 = self.smpl.lbs_hip(pred_shape, rotmat)
            extra9 = self.smpl.joint_regress_hip(self.smpl.J_regressor_extra, verts)
            jover, Jo = None, 0
            if J_regressor is not None:
                jover = self.smpl.joint_regress_hip(J_regressor.to(verts.device, torch.float32), verts)
                Jo = jover.shape[1]
            nj = Jo if jover is not None else 49
            kp3d = torch.empty(nt, nj, 3, dtype=torch.float32, device=verts.device)
            kp2d = torch.empty(nt, nj, 2, dtype=torch.float32, device=verts.device)
            ops.check(L.lib().maed_smpl_joints_project_fwd(ops._p(j24), ops._p(verts), ops._p(self.smpl.extra_vertex_ids), ops._p(extra9),
                                                           ops._p(self.smpl.joint_map), ops._p(pred_cam.contiguous()), ops._p(jover), Jo,
                                                           ops._p(kp3d), ops._p(kp2d), nt, ops._stream()), "smpl_joints_project_fwd")
            theta = torch.cat([pred_cam, aa, pred_shape], dim=1)
            return dict(theta=theta, verts=verts, kp_2d=kp2d, kp_3d=kp3d, rotmat=rotmat)
        rotmat = rot6d_to_rotmat(pred_pose).reshape(nt, -1, 3, 3)
        out = self.smpl(betas=pred_shape, body_pose=rotmat[:, 1:], global_orient=rotmat[:, 0].unsqueeze(1), pose2rot=False)
        verts, joints = out.vertices[:nt], out.joints[:nt]
        if J_regressor is not None:
            joints = self.smpl.regress_joints(J_regressor.to(verts.device, verts.dtype), verts)
        kp2d = projection(joints, pred_cam)
        aa = rotation_matrix_to_angle_axis(rotmat.reshape(-1, 3, 3)).reshape(nt, -1)
        return dict(theta=torch.cat([pred_cam, aa, pred_shape], dim=1), verts=verts, kp_2d=kp2d, kp_3d=joints, rotmat=rotmat)
