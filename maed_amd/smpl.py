"""SMPL body model wrapper (reference: lib/models/smpl.py on top of third-party smplx==0.1.13).

The reference subclasses smplx.SMPL and needs the licensed SMPL model file; neither exists here
(parity of the LBS arithmetic is UNPINNED).  This module restates the published
formulation and takes the model arrays from a dict / .npz / .pkl-derived
dict with the smplx field names; `SMPL.synthetic()` builds deterministic SMPL-shaped stand-in
parameters so the path can run and be timed without licensed data.

forward(betas, body_pose, global_orient, pose2rot=False) keeps smplx's call signature used at
lib/models/ktd.py:100-105 and returns an object with .vertices and .joints (49 joints through
joint_map, smpl.py:97-99).
"""
import ctypes as C
import os
import warnings
import weakref
from collections import namedtuple

import torch
import torch.nn as nn

from . import _lib as L
from . import ops

# lib/models/smpl.py:16-53,89
JOINT_MAP = {
    'OP Nose': 24, 'OP Neck': 12, 'OP RShoulder': 17, 'OP RElbow': 19, 'OP RWrist': 21, 'OP LShoulder': 16,
    'OP LElbow': 18, 'OP LWrist': 20, 'OP MidHip': 0, 'OP RHip': 2, 'OP RKnee': 5, 'OP RAnkle': 8,
    'OP LHip': 1, 'OP LKnee': 4, 'OP LAnkle': 7, 'OP REye': 25, 'OP LEye': 26, 'OP REar': 27,
    'OP LEar': 28, 'OP LBigToe': 29, 'OP LSmallToe': 30, 'OP LHeel': 31, 'OP RBigToe': 32, 'OP RSmallToe': 33,
    'OP RHeel': 34, 'Right Ankle': 8, 'Right Knee': 5, 'Right Hip': 45, 'Left Hip': 46, 'Left Knee': 4,
    'Left Ankle': 7, 'Right Wrist': 21, 'Right Elbow': 19, 'Right Shoulder': 17, 'Left Shoulder': 16,
    'Left Elbow': 18, 'Left Wrist': 20, 'Neck (LSP)': 47, 'Top of Head (LSP)': 48, 'Pelvis (MPII)': 49,
    'Thorax (MPII)': 50, 'Spine (H36M)': 51, 'Jaw (H36M)': 52, 'Head (H36M)': 53, 'Nose': 24, 'Left Eye': 26,
    'Right Eye': 25, 'Left Ear': 28, 'Right Ear': 27,
}
JOINT_NAMES = [
    'OP Nose', 'OP Neck', 'OP RShoulder', 'OP RElbow', 'OP RWrist', 'OP LShoulder', 'OP LElbow', 'OP LWrist',
    'OP MidHip', 'OP RHip', 'OP RKnee', 'OP RAnkle', 'OP LHip', 'OP LKnee', 'OP LAnkle', 'OP REye', 'OP LEye',
    'OP REar', 'OP LEar', 'OP LBigToe', 'OP LSmallToe', 'OP LHeel', 'OP RBigToe', 'OP RSmallToe', 'OP RHeel',
    'Right Ankle', 'Right Knee', 'Right Hip', 'Left Hip', 'Left Knee', 'Left Ankle', 'Right Wrist', 'Right Elbow',
    'Right Shoulder', 'Left Shoulder', 'Left Elbow', 'Left Wrist', 'Neck (LSP)', 'Top of Head (LSP)',
    'Pelvis (MPII)', 'Thorax (MPII)', 'Spine (H36M)', 'Jaw (H36M)', 'Head (H36M)', 'Nose', 'Left Eye',
    'Right Eye', 'Left Ear', 'Right Ear',
]
JOINT_IDS = {JOINT_NAMES[i]: i for i in range(len(JOINT_NAMES))}
H36M_TO_J17 = [6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 0, 7, 9, 10]
H36M_TO_J14 = [6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 10]

SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
# smplx vertex_ids['smpl'] in VertexJointSelector order (face, feet, finger tips)
SMPL_EXTRA_VERTEX_IDS = [332, 6260, 2800, 4071, 583, 3216, 3226, 3387, 6617, 6624, 6787,
                         2746, 2319, 2445, 2556, 2673, 6191, 5782, 5905, 6016, 6133]
N_VERTS = 6890

ModelOutput = namedtuple('ModelOutput', ['vertices', 'joints', 'full_pose', 'betas', 'global_orient', 'body_pose'])


def synthetic_smpl_arrays(seed=0):
    """Deterministic SMPL-shaped arrays (same generator as the test oracle's stand-in model)."""
    g = torch.Generator().manual_seed(seed)
    V = N_VERTS
    out = dict(v_template=torch.randn(V, 3, generator=g) * 0.3, shapedirs=torch.randn(V, 3, 10, generator=g) * 0.01,
               posedirs=torch.randn(207, V * 3, generator=g) * 0.001)

    def sparse_rows(rows, nnz):
        m = torch.zeros(rows, V)
        for r in range(rows):
            idx = torch.randperm(V, generator=g)[:nnz]
            w = torch.rand(nnz, generator=g) + 0.1
            m[r, idx] = w / w.sum()
        return m

    out['J_regressor'] = sparse_rows(24, 40)
    out['J_regressor_extra'] = sparse_rows(9, 30)
    out['J_regressor_h36m'] = sparse_rows(17, 50)
    wj = torch.randint(0, 24, (V, 4), generator=g)
    ww = torch.rand(V, 4, generator=g) + 0.05
    lw = torch.zeros(V, 24)
    lw.scatter_add_(1, wj, ww)
    out['lbs_weights'] = lw / lw.sum(1, keepdim=True)
    return out


class SMPL(nn.Module):
    """Buffers use smplx's names so `decoder.smpl.*` state_dict entries line up (they are filtered
    out on load by the reference anyway: eval.py:29, train.py:101)."""

    def __init__(self, model_arrays=None, **kwargs):
        super().__init__()
        a = model_arrays if model_arrays is not None else synthetic_smpl_arrays(0)
        self.synthetic = model_arrays is None
        if self.synthetic and os.environ.get("MAED_SYNTHETIC_SMPL_OK") != "1":
            warnings.warn("maed_amd.SMPL: no model arrays given -- using the deterministic synthetic stand-in (SMPL-shaped random parameters). "
                          "Pass smpl_arrays= (or load a state_dict that carries decoder.smpl.*) for real meshes; set MAED_SYNTHETIC_SMPL_OK=1 "
                          "to silence this in tests and benchmarks.", stacklevel=3)
        f = lambda k: torch.as_tensor(a[k], dtype=torch.float32).contiguous()
        self.register_buffer('v_template', f('v_template'))
        self.register_buffer('shapedirs', f('shapedirs')[:, :, :10].contiguous())
        self.register_buffer('posedirs', f('posedirs').reshape(207, N_VERTS * 3).contiguous())
        self.register_buffer('J_regressor', f('J_regressor'))
        self.register_buffer('lbs_weights', f('lbs_weights'))
        self.register_buffer('J_regressor_extra', f('J_regressor_extra'))
        self.register_buffer('parents', torch.tensor(SMPL_PARENTS, dtype=torch.int32), persistent=False)
        self.register_buffer('extra_vertex_ids', torch.tensor(SMPL_EXTRA_VERTEX_IDS, dtype=torch.long), persistent=False)
        self.register_buffer('joint_map', torch.tensor([JOINT_MAP[n] for n in JOINT_NAMES], dtype=torch.long), persistent=False)
        # rest-pose joint regression folded once: J = J_regressor (v_template + shapedirs beta).  DERIVED from the persistent buffers:
        # refreshed whenever those change (load_state_dict of a reference checkpoint carrying decoder.smpl.*, in-place edits, .to())
        self.register_buffer('J_template', self.J_regressor @ self.v_template, persistent=False)
        self.register_buffer('J_shapedirs', torch.einsum('jv,vcl->jcl', self.J_regressor, self.shapedirs).contiguous(), persistent=False)
        self.faces = None
        self._ps_t = None
        self._derived_key = self._base_key()
        # `synthetic` is cleared only when a state_dict really delivered the model arrays (not by device moves or in-place version bumps: ADVICE r2)
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._after_load(incompatible))

    def _after_load(self, incompatible):
        base = ("v_template", "shapedirs", "posedirs", "J_regressor")
        missing = {k.rsplit(".", 1)[-1] for k in incompatible.missing_keys}
        self.refresh_derived(force=True)
        if not missing.intersection(base):      # the state_dict carried the model arrays (the reference's loaders filter decoder.smpl.* out: eval.py:29)
            self.synthetic = False

    def _base_key(self):
        return tuple((t.data_ptr(), t._version) for t in (self.v_template, self.shapedirs, self.posedirs, self.J_regressor))

    def refresh_derived(self, force=False):
        """J_template, J_shapedirs and the backward's GEMM operand follow v_template / shapedirs / posedirs / J_regressor"""
        key = self._base_key()
        changed = key != self._derived_key
        if force or changed:
            with torch.no_grad():
                self.J_template = self.J_regressor @ self.v_template
                self.J_shapedirs = torch.einsum('jv,vcl->jcl', self.J_regressor, self.shapedirs).contiguous()
            self._ps_t = None
            self._derived_key = self._base_key()

    def pose_shape_dirs(self):
        """(207+10, 20670) = [posedirs ; shapedirs^T]: the K-contiguous B operand of the LBS backward's GEMM (tail.SmplTailFn)"""
        self.refresh_derived()
        if self._ps_t is None or self._ps_t.device != self.posedirs.device:
            self._ps_t = torch.cat([self.posedirs, self.shapedirs.reshape(-1, 10).t()], dim=0).contiguous()
        return self._ps_t

    def active_vertices(self):
        """sorted int32 ids of the vertices the 9 regressed + 21 selected extra joints read (non-zero columns of J_regressor_extra, extra_vertex_ids): where the
        vertex gradient of the training objective can be non-zero (tail.SmplTailFn backward without d_verts).  Derived once per regressor state (one host sync)."""
        key = (self.J_regressor_extra.data_ptr(), self.J_regressor_extra._version, str(self.J_regressor_extra.device))
        if getattr(self, "_active_key", None) != key:
            with torch.no_grad():
                cols = (self.J_regressor_extra != 0).any(dim=0)
                cols[self.extra_vertex_ids.to(cols.device)] = True
                self._active = torch.nonzero(cols).flatten().to(torch.int32).contiguous()
            self._active_key, self._ps_active = key, None
        return self._active

    def pose_shape_dirs_active(self):
        """(207+10, 3 * n_active): the columns of pose_shape_dirs() that belong to active_vertices()"""
        act = self.active_vertices()
        full = self.pose_shape_dirs()
        if getattr(self, "_ps_active", None) is None or self._ps_active_src is not full:
            cols = (act.long()[:, None] * 3 + torch.arange(3, device=act.device)[None, :]).flatten()
            self._ps_active, self._ps_active_src = full[:, cols].contiguous(), full
        return self._ps_active

    def pose_shape_dirs_t(self):
        """(20670, 207+10) view of pose_shape_dirs()"""
        return self.pose_shape_dirs().t()

    # ---- ATen path (training graph; differentiable) ------------------------------------------------
    # Written for the GPU: every contraction is ONE large GEMM (frames folded into the N dimension) or a
    # broadcast multiply-sum; no batched 3x3 / 4x4 matmuls (882k-batch bmm's were 25 ms of a 94 ms step)
    # and the kinematic chain advances one tree LEVEL (9) at a time instead of one joint (24) at a time.
    _LEVELS = [[0], [1, 2, 3], [4, 5, 6], [7, 8, 9], [10, 11, 12, 13, 14], [15, 16, 17], [18, 19], [20, 21], [22, 23]]

    def lbs_torch(self, betas, rotmat):
        self.refresh_derived()
        Fr = betas.shape[0]
        v_shaped = self.v_template.reshape(1, -1) + betas @ self.shapedirs.reshape(-1, 10).t()          # (F, 20670)
        J = (self.J_template.reshape(1, -1) + betas @ self.J_shapedirs.reshape(-1, 10).t()).reshape(Fr, 24, 3)
        ident = torch.eye(3, dtype=betas.dtype, device=betas.device)
        pose_feature = (rotmat[:, 1:] - ident).reshape(Fr, 207)
        v_posed = (v_shaped + pose_feature @ self.posedirs).reshape(Fr, -1, 3)
        Rw, tw = [None] * 24, [None] * 24
        Rw[0], tw[0] = rotmat[:, 0], J[:, 0]
        for level in self._LEVELS[1:]:
            par = [SMPL_PARENTS[j] for j in level]
            lo, hi = level[0], level[-1] + 1                 # every level is a contiguous joint range: plain slices,
            Rp = torch.stack([Rw[p] for p in par], 1)        # no host-built index tensors (hipGraph-capturable)   (F,k,3,3)
            tp = torch.stack([tw[p] for p in par], 1)
            Jp = torch.stack([J[:, p] for p in par], 1)
            Rn = (Rp.unsqueeze(-1) * rotmat[:, lo:hi].unsqueeze(-3)).sum(-2)                             # Rp @ R_j
            tn = (Rp * (J[:, lo:hi] - Jp).unsqueeze(-2)).sum(-1) + tp                                    # Rp @ rel + tp
            for i, j in enumerate(level):
                Rw[j], tw[j] = Rn[:, i], tn[:, i]
        Rw, tw = torch.stack(Rw, 1), torch.stack(tw, 1)
        trel = tw - (Rw * J.unsqueeze(-2)).sum(-1)
        A = torch.cat([Rw, trel.unsqueeze(-1)], dim=-1).reshape(Fr, 24, 12)
        Tv = (self.lbs_weights @ A.permute(1, 0, 2).reshape(24, Fr * 12)).reshape(-1, Fr, 3, 4).permute(1, 0, 2, 3)  # (F,V,3,4)
        verts = (Tv[..., :3] * v_posed.unsqueeze(-2)).sum(-1) + Tv[..., 3]
        return verts, tw

    @staticmethod
    def regress_joints(Jreg, verts):
        """(J,V) x (F,V,3) -> (F,J,3) as ONE GEMM with the frames folded into N"""
        Fr = verts.shape[0]
        return (Jreg @ verts.permute(1, 0, 2).reshape(verts.shape[1], Fr * 3)).reshape(-1, Fr, 3).permute(1, 0, 2)

    def joints49_torch(self, verts, joints24):
        j45 = torch.cat([joints24, verts[:, self.extra_vertex_ids]], dim=1)
        extra = self.regress_joints(self.J_regressor_extra, verts)
        return torch.cat([j45, extra], dim=1)[:, self.joint_map]

    # ---- HIP path (inference) ----------------------------------------------------------------------------
    def _c_params(self):
        self.refresh_derived()
        sp = L.SmplParams()
        for k in ['v_template', 'shapedirs', 'posedirs', 'J_template', 'J_shapedirs', 'lbs_weights', 'parents']:
            setattr(sp, k, getattr(self, k).data_ptr())
        return sp

    def lbs_hip(self, betas, rotmat):
        Fr = betas.shape[0]
        dev = betas.device
        verts = torch.empty(Fr, N_VERTS, 3, dtype=torch.float32, device=dev)
        j24 = torch.empty(Fr, 24, 3, dtype=torch.float32, device=dev)
        A = torch.empty(Fr, 24, 12, dtype=torch.float32, device=dev)
        sp = self._c_params()
        ops.check(L.lib().maed_smpl_lbs_fwd(C.byref(sp), ops._p(betas.contiguous()), ops._p(rotmat.contiguous()), ops._p(verts),
                                            ops._p(j24), ops._p(A), None, Fr, ops._stream()), 'smpl_lbs_fwd')
        return verts, j24

    def regressor_csr(self, Jreg):
        """(rowptr, cols, vals) int32 / int32 / fp32 device tensors of a (J, 6890) joint regressor, or None when it is not sparse enough to bother (> 10 % non-zero)
        or has more than 64 rows.  Built once per regressor tensor and version (one host sync)."""
        # keyed by the tensor OBJECT (weak reference) + its version: a storage address alone can be handed to a different regressor after the first one is freed
        cache = self.__dict__.setdefault("_csr_cache", [])
        cache[:] = [e for e in cache if e[0]() is not None][-8:]
        for ref, version, csr in cache:
            if ref() is Jreg and version == Jreg._version:
                return csr
        with torch.no_grad():
            nz = Jreg != 0
            if Jreg.dim() != 2 or Jreg.shape[0] > 64 or Jreg.shape[1] != N_VERTS or float(nz.float().mean()) > 0.10 or os.environ.get("MAED_JREG_CSR", "1") == "0":
                csr = None
            else:
                idx = torch.nonzero(nz)                  # row-major: ascending vertex id inside a row
                counts = torch.bincount(idx[:, 0], minlength=Jreg.shape[0])
                rowptr = torch.zeros(Jreg.shape[0] + 1, dtype=torch.int32, device=Jreg.device)
                rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
                csr = (rowptr, idx[:, 1].to(torch.int32).contiguous(), Jreg[nz].float().contiguous())
        cache.append((weakref.ref(Jreg), Jreg._version, csr))
        return csr

    def joint_regress_hip(self, Jreg, verts):
        Fr, J = verts.shape[0], Jreg.shape[0]
        out = torch.empty(Fr, J, 3, dtype=torch.float32, device=verts.device)
        csr = self.regressor_csr(Jreg)
        if csr is not None:
            ops.check(L.lib().maed_joint_regress_csr_fwd(ops._p(csr[0]), ops._p(csr[1]), ops._p(csr[2]), J, ops._p(verts), ops._p(out), Fr, ops._stream()), 'joint_regress_csr_fwd')
            return out
        ops.check(L.lib().maed_joint_regress_fwd(ops._p(Jreg.contiguous()), J, ops._p(verts), ops._p(out), Fr, ops._stream()), 'joint_regress_fwd')
        return out

    def forward(self, betas=None, body_pose=None, global_orient=None, pose2rot=False, **kwargs):
        if pose2rot:
            raise NotImplementedError('pose2rot=True is not used on the MAED path (ktd.py:104)')
        rot = torch.cat([global_orient, body_pose], dim=1)
        needs_grad = torch.is_grad_enabled() and (betas.requires_grad or rot.requires_grad)
        if not ops.on_library_device(betas):                # host tensors: the ATen composition (device-agnostic, differentiable)
            verts, j24 = self.lbs_torch(betas, rot)
            joints = self.joints49_torch(verts, j24)
        elif needs_grad:                                    # differentiable on the library: the decoder tail's kernels behind one Function
            from . import tail
            verts, joints = tail.SmplLbsFn.apply(betas, rot, self)
        else:
            verts, j24 = self.lbs_hip(betas.float(), rot.float())
            extra = self.joint_regress_hip(self.J_regressor_extra, verts)
            j54 = torch.cat([j24, verts[:, self.extra_vertex_ids], extra], dim=1)
            joints = j54[:, self.joint_map]
        return ModelOutput(vertices=verts, joints=joints, full_pose=rot, betas=betas, global_orient=global_orient, body_pose=body_pose)
