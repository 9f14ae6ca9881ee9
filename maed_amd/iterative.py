"""Iterative (SPIN/HMR-style) SMPL regressor, decoder='iterative' (reference: lib/models/spin.py:17-110).

Same module / parameter / buffer names as the reference (`fc1, drop1, fc2, drop2, decpose, decshape, deccam,
init_pose, init_shape, init_cam, smpl`) and the same forward contract as KTD; everything downstream of the
(pose6d, shape, cam) triple is the KTD tail (maed_amd/ktd.py get_output, tail.SmplTailFn), so the SMPL / projection
kernels and their hand-written backward are shared.

One restructuring against the reference's loop (spin.py:67-74): fc1 acts on cat([x, pose, shape, cam]) three times
with the same x, so the (F, feat_dim) x (feat_dim, hidden) product -- 93% of fc1's work at feat_dim 2048 -- is
computed once and only the 157-column parameter part is recomputed per iteration:
    fc1(cat[x, p]) = x W[:, :feat]^T + b  +  p W[:, feat:]^T.
Eval / no-grad runs on libmaed_hip.so's exact-f32 GEMM (strided views of fc1.weight, no copies); training keeps the
four small GEMMs + Dropout in ATen like KTD's fc1/fc2 and enters HIP at tail.SmplTailFn.

`smpl_mean_params`: None (deterministic stand-in: identity rotations, zero betas, camera (0.9, 0, 0) -- the SPIN
data file is not redistributable), a path to smpl_mean_params.npz, or a dict with 'pose' (144,), 'shape' (10,), 'cam' (3,).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import ops
from . import tail
from .ktd import KTD
from .smpl import SMPL

NPOSE, NSHAPE, NCAM = 24 * 6, 10, 3


def _mean_params(src):
    if src is None:
        return dict(pose=torch.tensor([1.0, 0.0, 0.0, 1.0, 0.0, 0.0]).repeat(24), shape=torch.zeros(NSHAPE),
                    cam=torch.tensor([0.9, 0.0, 0.0]))
    if isinstance(src, (str, bytes)):
        src = np.load(src)
    return {k: torch.as_tensor(np.asarray(src[k]), dtype=torch.float32).reshape(-1) for k in ('pose', 'shape', 'cam')}


class Regressor(nn.Module):
    def __init__(self, smpl_mean_params=None, feat_dim=2048, hidden_dim=1024, smpl_arrays=None, **kwargs):
        super().__init__()
        self.feat_dim, self.hidden_dim = feat_dim, hidden_dim
        self.smpl = SMPL(smpl_arrays)
        self.fc1 = nn.Linear(feat_dim + NPOSE + NSHAPE + NCAM, hidden_dim)
        self.drop1 = nn.Dropout()
        self.fc2 = nn.Linear(hidden_dim, hidden_dim)
        self.drop2 = nn.Dropout()
        self.decpose = nn.Linear(hidden_dim, NPOSE)
        self.decshape = nn.Linear(hidden_dim, NSHAPE)
        self.deccam = nn.Linear(hidden_dim, NCAM)
        for m in (self.decpose, self.decshape, self.deccam):
            nn.init.xavier_uniform_(m.weight, gain=0.01)
        mp = _mean_params(smpl_mean_params)
        self.register_buffer('init_pose', mp['pose'].reshape(1, NPOSE))
        self.register_buffer('init_shape', mp['shape'].reshape(1, NSHAPE))
        self.register_buffer('init_cam', mp['cam'].reshape(1, NCAM))
        self._packed_key, self._packed = None, None
        # fp32 [out,in] masters + transposed images for the training head's library GEMMs (one cache per weight view: WeightCache keys on the tensors it is given)
        self._caches = {k: ops.WeightCache() for k in ("fc1x", "fc1p", "fc2", "pose", "shape", "cam")}

    # the tail and its dispatch predicates are KTD's (same attributes: fc1, smpl, training)
    get_output = KTD.get_output
    _use_hip = KTD._use_hip
    _use_hip_train = KTD._use_hip_train

    def _decoders(self):
        """decpose | decshape | deccam as one (157, hidden) weight: one GEMM per iteration instead of three"""
        key = (ops.WEIGHT_EPOCH, tuple(p._version for p in self.parameters()), self.decpose.weight.data_ptr())
        if key != self._packed_key:
            with torch.no_grad():
                w = torch.cat([self.decpose.weight, self.decshape.weight, self.deccam.weight], 0).contiguous()
                b = torch.cat([self.decpose.bias, self.decshape.bias, self.deccam.bias], 0).contiguous()
            self._packed_key, self._packed = key, (w, b)
        return self._packed

    def _init(self, nt, init_pose, init_shape, init_cam):
        pose = self.init_pose.expand(nt, -1) if init_pose is None else init_pose
        shape = self.init_shape.expand(nt, -1) if init_shape is None else init_shape
        cam = self.init_cam.expand(nt, -1) if init_cam is None else init_cam
        return pose, shape, cam

    def _regress_hip(self, x, pose, shape, cam, n_iter):
        fd = self.feat_dim
        w1 = self.fc1.weight.detach()
        w_dec, b_dec = self._decoders()
        x = x.float().contiguous()
        hx = ops.gemm_nt(x, w1[:, :fd], L.EPI_STORE, bias=self.fc1.bias)            # iteration-invariant part of fc1
        prm = torch.cat([pose, shape, cam], dim=1).float().contiguous()             # (F, 157)
        for _ in range(n_iter):
            h1 = ops.gemm_nt(prm, w1[:, fd:], L.EPI_STORE).add_(hx)
            h2 = ops.gemm_nt(h1, self.fc2.weight.detach(), L.EPI_STORE, bias=self.fc2.bias)
            prm = ops.gemm_nt(h2, w_dec, L.EPI_STORE, bias=b_dec).add_(prm)
        return prm[:, :NPOSE].contiguous(), prm[:, NPOSE:NPOSE + NSHAPE].contiguous(), prm[:, NPOSE + NSHAPE:].contiguous()

    def _regress_train(self, x, pose, shape, cam, n_iter):
        """the differentiable head on the library (spin.py:53-76): every nn.Linear through ste_modes.LinearTokFn (maed_gemm_nt forward / input gradient,
        maed_gemm_tn_wgrad or transposed copies for the weight gradient), Dropout through maed_dropout; fc1's iteration-invariant feature part hoisted
        out of the three rounds as in _regress_hip.  What stays on ATen are the (F,157) adds and concatenations between the GEMMs."""
        from . import ste_modes
        lin = ste_modes.LinearTokFn.apply
        fd, c = self.feat_dim, self._caches
        x = x.float().contiguous()
        w1 = self.fc1.weight
        hx = lin(x, w1[:, :fd], self.fc1.bias, c["fc1x"], True)
        for _ in range(n_iter):
            prm = torch.cat([pose, shape, cam], dim=1).float().contiguous()
            xc = ste_modes.dropout(hx + lin(prm, w1[:, fd:], None, c["fc1p"], True), self.drop1.p, self.drop1.training)
            xc = ste_modes.dropout(lin(xc, self.fc2.weight, self.fc2.bias, c["fc2"], True), self.drop2.p, self.drop2.training)
            pose = lin(xc, self.decpose.weight, self.decpose.bias, c["pose"], True) + pose
            shape = lin(xc, self.decshape.weight, self.decshape.bias, c["shape"], True) + shape
            cam = lin(xc, self.deccam.weight, self.deccam.bias, c["cam"], True) + cam
        return pose, shape, cam

    def _regress_torch(self, x, pose, shape, cam, n_iter):
        fd = self.feat_dim
        hx = F.linear(x, self.fc1.weight[:, :fd], self.fc1.bias)
        w_p = self.fc1.weight[:, fd:]
        for _ in range(n_iter):
            xc = self.drop1(hx + F.linear(torch.cat([pose, shape, cam], dim=1), w_p))
            xc = self.drop2(self.fc2(xc))
            pose = self.decpose(xc) + pose
            shape = self.decshape(xc) + shape
            cam = self.deccam(xc) + cam
        return pose, shape, cam

    def iterative_regress(self, x, init_pose=None, init_shape=None, init_cam=None, n_iter=3):
        """spin.py:53-76"""
        pose, shape, cam = self._init(x.shape[0], init_pose, init_shape, init_cam)
        if self._use_hip(x):
            return self._regress_hip(x, pose, shape, cam, n_iter)
        if ops.on_library_device(x):                 # differentiable, on the device: library Functions (no ATen twin on a GPU)
            return self._regress_train(x, pose, shape, cam, n_iter)
        return self._regress_torch(x.float(), pose, shape, cam, n_iter)     # host tensors: the ATen composition the CPU suite compares the kernels with

    def forward(self, x, seqlen, J_regressor=None, init_pose=None, init_shape=None, init_cam=None, n_iter=3, **kwargs):
        """spin.py:78-86.  The reference ignores its n_iter argument here and always runs 3 rounds (spin.py:83); so does this."""
        hip = self._use_hip(x)
        hip_train = self._use_hip_train(x, J_regressor)
        pred_pose, pred_shape, pred_cam = self.iterative_regress(x, init_pose, init_shape, init_cam, n_iter=3)
        if hip_train:
            theta, verts, kp2d, kp3d, rotmat = tail.SmplTailFn.apply(pred_pose, pred_shape, pred_cam, self.smpl)
            return dict(theta=theta, verts=verts, kp_2d=kp2d, kp_3d=kp3d, rotmat=rotmat)
        return self.get_output(pred_pose, pred_shape, pred_cam, J_regressor, hip)
