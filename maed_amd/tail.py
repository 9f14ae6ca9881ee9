"""Decoder tail of the TRAINING graph on libmaed_hip (reference: lib/models/ktd.py:69-124).

Two autograd Functions whose forward is the same HIP code the inference path runs and whose backward is
~10 hand-written launches (maed_amd/csrc/tail_bwd.hip) instead of the ~1000 tiny ATen kernels autograd
needs for the 24-joint chains:

  KtdChainFn : h2 (F,hidden) -> pose6d (F,144), shape (F,10), cam (F,3)      ktd.py:78-86
               one packed (157 x hidden) GEMM + the serial ancestor chain; the 52 regressor tensors get their
               gradients written straight into .grad (same protocol as the STE blocks / backbone: the owner's
               `grads_ready` callback tells the data-parallel bucketer when they are final).
  SmplTailFn : pose6d, shape, cam -> theta, verts, kp_2d, kp_3d, rotmat      ktd.py:94-124

Every GEMM of the two Functions -- the packed (157 x hidden) projection and its two backward products, and the (F,20670) x (20670,217)
pose-/shape-direction product of the LBS backward -- runs on maed_gemm_nt (exact-fp32 kernel, split-K where the output is tiny).
"""
import ctypes as C
import os

import torch

from . import _lib as L
from . import ops

N_VERTS = 6890


def _f32(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _off(t, elems):
    """device pointer `elems` floats into a contiguous fp32 tensor (None stays None)"""
    return None if t is None else ops._p(t) + 4 * elems


class KtdChainFn(ops.ReportingFn):
    @staticmethod
    def forward(ctx, h2, ktd, *params):
        lib = L.lib()
        h2 = _f32(h2)
        Fr, hidden = h2.shape
        dev = h2.device
        w_feat = torch.empty(157, hidden, dtype=torch.float32, device=dev)
        b_feat = torch.empty(157, dtype=torch.float32, device=dev)
        w_anc = torch.empty(L.KTD_W_ANC, dtype=torch.float32, device=dev)
        tbl = ktd._ptr_table(False)
        L.check(lib.maed_ktd_pack(C.byref(tbl), hidden, ops._p(w_feat), ops._p(b_feat), ops._p(w_anc), ops._stream()), "ktd_pack")
        hm = getattr(ktd, "head_matmul", None)
        out = ops.gemm_nt(h2, w_feat, L.EPI_STORE, bias=b_feat, prec=hm)             # (F, 144 + 10 + 3)
        base = out[:, :144].contiguous()
        pose = torch.empty_like(base)
        L.check(lib.maed_ktd_chain_fwd(ops._p(base), ops._p(w_anc), ops._p(pose), Fr, ops._stream()), "ktd_chain_fwd")
        ctx.save_for_backward(h2, pose, w_feat, w_anc)
        ctx.ktd, ctx.hm = ktd, hm
        ctx.set_materialize_grads(False)
        if ops.ReportingFn.will_run_backward(ctx):
            ktd._pending_backwards += 1
        return pose, out[:, 144:154].contiguous(), out[:, 154:157].contiguous()

    @staticmethod
    def backward(ctx, d_pose, d_shape, d_cam):
        lib = L.lib()
        h2, pose, w_feat, w_anc = ctx.saved_tensors
        ktd = ctx.ktd
        Fr, hidden = h2.shape
        dev = h2.device
        d_pose = torch.zeros_like(pose) if d_pose is None else _f32(d_pose)
        d_shape = None if d_shape is None else _f32(d_shape)
        d_cam = None if d_cam is None else _f32(d_cam)
        d_out = torch.empty(Fr, 157, dtype=torch.float32, device=dev)
        d_w_anc = torch.empty(L.KTD_W_ANC, dtype=torch.float32, device=dev)
        d_b = torch.empty(157, dtype=torch.float32, device=dev)
        L.check(lib.maed_ktd_chain_bwd(ops._p(pose), ops._p(w_anc), ops._p(d_pose), ops._p(d_shape), ops._p(d_cam), ops._p(d_out), 157,
                                       ops._p(d_w_anc), ops._p(d_b), Fr, ops._stream()), "ktd_chain_bwd")
        # d_h2 = d_out W (K = 157), d_W = d_out^T h2 (K = F): NT GEMMs on transposed copies of the three small operands
        d_out_t, _ = ops.transpose_cast(d_out, torch.float32, pad_to=1)             # (157, F)
        d_h2 = None
        if ctx.needs_input_grad[0]:
            w_feat_t, _ = ops.transpose_cast(w_feat, torch.float32, pad_to=1)       # (hidden, 157)
            d_h2 = ops.gemm_nt(d_out, w_feat_t, L.EPI_STORE, prec=ctx.hm)           # (K = 157 is not a multiple of 32: the exact kernel either way)
        h2_t, _ = ops.transpose_cast(h2, torch.float32, pad_to=1)                   # (hidden, F)
        d_w_feat = ops.gemm_nt(d_out_t, h2_t, L.EPI_STORE, prec=ctx.hm)             # (157, hidden)
        tbl = ktd._ptr_table(True)
        L.check(lib.maed_ktd_unpack_add(C.byref(tbl), hidden, ops._p(d_w_feat), ops._p(d_b), ops._p(d_w_anc), ops._stream()), "ktd_unpack_add")
        ktd._pending_backwards -= 1
        if ktd._pending_backwards == 0 and ktd.grads_ready is not None:
            ktd.grads_ready(ktd)
        return (d_h2, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class SmplTailFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pose6d, shape, cam, smpl):
        lib = L.lib()
        pose6d, shape, cam = _f32(pose6d), _f32(shape), _f32(cam)
        Fr = pose6d.shape[0]
        dev = pose6d.device
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        rotmat, aa = new(Fr, 24, 3, 3), new(Fr, 72)
        L.check(lib.maed_rot6d_pose_fwd(ops._p(pose6d), ops._p(rotmat), ops._p(aa), Fr * 24, ops._stream()), "rot6d_pose_fwd")
        verts, j24, A, v_posed = new(Fr, N_VERTS, 3), new(Fr, 24, 3), new(Fr, 24, 12), new(Fr, N_VERTS, 3)
        sp = smpl._c_params()
        L.check(lib.maed_smpl_lbs_fwd(C.byref(sp), ops._p(shape), ops._p(rotmat), ops._p(verts), ops._p(j24), ops._p(A), ops._p(v_posed), Fr,
                                      ops._stream()), "smpl_lbs_fwd")
        extra9 = smpl.joint_regress_hip(smpl.J_regressor_extra, verts)
        kp3d, kp2d = new(Fr, 49, 3), new(Fr, 49, 2)
        L.check(lib.maed_smpl_joints_project_fwd(ops._p(j24), ops._p(verts), ops._p(smpl.extra_vertex_ids), ops._p(extra9), ops._p(smpl.joint_map),
                                                 ops._p(cam), None, 0, ops._p(kp3d), ops._p(kp2d), Fr, ops._stream()), "smpl_joints_project_fwd")
        theta = torch.cat([cam, aa, shape], dim=1)
        ctx.save_for_backward(pose6d, shape, cam, rotmat, A, v_posed, kp3d)
        ctx.smpl = smpl
        ctx.set_materialize_grads(False)
        return theta, verts, kp2d, kp3d, rotmat

    @staticmethod
    def backward(ctx, d_theta, d_verts, d_kp2d, d_kp3d, d_rotmat):
        lib = L.lib()
        pose6d, shape, cam, rotmat, A, v_posed, kp3d = ctx.saved_tensors
        smpl = ctx.smpl
        Fr = pose6d.shape[0]
        dev = pose6d.device
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        d_theta, d_verts, d_kp2d, d_kp3d, d_rotmat = [None if g is None else _f32(g) for g in (d_theta, d_verts, d_kp2d, d_kp3d, d_rotmat)]
        st = ops._stream()
        d_j24, d_e21, d_e9, d_cam = new(Fr, 24, 3), new(Fr, 21, 3), new(Fr, 9, 3), new(Fr, 3)
        L.check(lib.maed_smpl_joints_project_bwd(ops._p(kp3d), ops._p(cam), ops._p(smpl.joint_map), ops._p(d_kp3d), ops._p(d_kp2d),
                                                 _off(d_theta, 0), 85, ops._p(d_j24), ops._p(d_e21), ops._p(d_e9), ops._p(d_cam), Fr, st),
                "smpl_joints_project_bwd")
        sp = smpl._c_params()
        dA = torch.zeros(Fr, 24, 12, dtype=torch.float32, device=dev)
        dpf = _smpl_vertex_backward(lib, smpl, sp, A, v_posed, d_verts, d_e21, d_e9, dA, Fr, st)
        d_rot, d_betas = new(Fr, 24, 9), new(Fr, 10)
        L.check(lib.maed_smpl_chain_bwd(C.byref(sp), ops._p(shape), ops._p(rotmat), ops._p(dA), ops._p(d_j24), ops._p(dpf), ops._p(d_rotmat),
                                        _off(d_theta, 75), 85, ops._p(d_rot), ops._p(d_betas), Fr, st), "smpl_chain_bwd")
        d_pose6d = new(Fr, 144)
        L.check(lib.maed_rot6d_pose_bwd(ops._p(pose6d), ops._p(d_rot), _off(d_theta, 3), 85, ops._p(d_pose6d), Fr * 24, st), "rot6d_pose_bwd")
        return d_pose6d, d_betas, d_cam, None


def _smpl_vertex_backward(lib, smpl, sp, A, v_posed, d_verts, d_e21, d_e9, dA, Fr, st):
    """vertex part of the LBS backward: dA (accumulated) and dpf (F, 207 + 10) = d_vposed [posedirs; shapedirs^T]^T.
    With a gradient on the vertices: every vertex (maed_smpl_skin_bwd) and the (F, 20670) x (20670, 217) product -- tiny output, split-K with the fp32 atomic
    epilogue.  Without (the training objective reads key points only): the ~290 vertices the extra joints read (maed_smpl_skin_bwd_sparse) and their ~870 columns
    (MAED_SMPL_SPARSE_BWD=0: the dense form either way, A/B knob)."""
    dev = A.device
    if d_verts is None and os.environ.get("MAED_SMPL_SPARSE_BWD", "1") == "1":
        act = smpl.active_vertices()
        na = act.numel()
        d_vp = torch.empty(Fr, na * 3, dtype=torch.float32, device=dev)
        L.check(lib.maed_smpl_skin_bwd_sparse(C.byref(sp), ops._p(A), ops._p(v_posed), ops._p(act), na, ops._p(d_e21), ops._p(smpl.extra_vertex_ids),
                                              ops._p(d_e9), ops._p(smpl.J_regressor_extra), ops._p(d_vp), ops._p(dA), Fr, st), "smpl_skin_bwd_sparse")
        return ops.gemm_nt(d_vp, smpl.pose_shape_dirs_active(), L.EPI_ATOMIC_F32, splitk=4)
    d_vposed = torch.empty(Fr, N_VERTS * 3, dtype=torch.float32, device=dev)
    L.check(lib.maed_smpl_skin_bwd(C.byref(sp), ops._p(A), ops._p(v_posed), ops._p(d_verts), ops._p(d_e21), ops._p(smpl.extra_vertex_ids),
                                   ops._p(d_e9), ops._p(smpl.J_regressor_extra), ops._p(d_vposed), ops._p(dA), Fr, st), "smpl_skin_bwd")
    return ops.gemm_nt(d_vposed, smpl.pose_shape_dirs(), L.EPI_ATOMIC_F32, splitk=32)


class SmplLbsFn(torch.autograd.Function):
    """SMPL.forward (lib/models/smpl.py:94-106 on smplx's lbs) as a differentiable stand-alone op on the library: betas (F,10), rotation
    matrices (F,24,3,3) -> vertices (F,6890,3), the 49 joints (joint_map gather, bit-exact).  Same kernels as SmplTailFn without its
    rot6d / projection ends (the projection kernels run with a unit camera and no 2D gradient)."""

    @staticmethod
    def forward(ctx, betas, rotmat, smpl):
        lib = L.lib()
        betas, rotmat = _f32(betas), _f32(rotmat)
        Fr, dev = betas.shape[0], betas.device
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        verts, j24, A, v_posed = new(Fr, N_VERTS, 3), new(Fr, 24, 3), new(Fr, 24, 12), new(Fr, N_VERTS, 3)
        sp = smpl._c_params()
        L.check(lib.maed_smpl_lbs_fwd(C.byref(sp), ops._p(betas), ops._p(rotmat), ops._p(verts), ops._p(j24), ops._p(A), ops._p(v_posed), Fr,
                                      ops._stream()), "smpl_lbs_fwd")
        extra9 = smpl.joint_regress_hip(smpl.J_regressor_extra, verts)
        cam = torch.zeros(Fr, 3, dtype=torch.float32, device=dev)
        cam[:, 0] = 1.0
        joints, kp2d = new(Fr, 49, 3), new(Fr, 49, 2)
        L.check(lib.maed_smpl_joints_project_fwd(ops._p(j24), ops._p(verts), ops._p(smpl.extra_vertex_ids), ops._p(extra9), ops._p(smpl.joint_map),
                                                 ops._p(cam), None, 0, ops._p(joints), ops._p(kp2d), Fr, ops._stream()), "smpl_joints_project_fwd")
        ctx.save_for_backward(betas, rotmat, A, v_posed, joints, cam)
        ctx.smpl = smpl
        ctx.set_materialize_grads(False)
        return verts, joints

    @staticmethod
    def backward(ctx, d_verts, d_joints):
        lib = L.lib()
        betas, rotmat, A, v_posed, joints, cam = ctx.saved_tensors
        smpl = ctx.smpl
        Fr, dev = betas.shape[0], betas.device
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        d_verts = None if d_verts is None else _f32(d_verts)
        d_joints = None if d_joints is None else _f32(d_joints)
        st = ops._stream()
        d_j24, d_e21, d_e9, d_cam = new(Fr, 24, 3), new(Fr, 21, 3), new(Fr, 9, 3), new(Fr, 3)
        L.check(lib.maed_smpl_joints_project_bwd(ops._p(joints), ops._p(cam), ops._p(smpl.joint_map), ops._p(d_joints), None, None, 0, ops._p(d_j24),
                                                 ops._p(d_e21), ops._p(d_e9), ops._p(d_cam), Fr, st), "smpl_joints_project_bwd")
        sp = smpl._c_params()
        dA = torch.zeros(Fr, 24, 12, dtype=torch.float32, device=dev)
        dpf = _smpl_vertex_backward(lib, smpl, sp, A, v_posed, d_verts, d_e21, d_e9, dA, Fr, st)
        d_rot, d_betas = new(Fr, 24, 9), new(Fr, 10)
        L.check(lib.maed_smpl_chain_bwd(C.byref(sp), ops._p(betas), ops._p(rotmat), ops._p(dA), ops._p(d_j24), ops._p(dpf), None, None, 0,
                                        ops._p(d_rot), ops._p(d_betas), Fr, st), "smpl_chain_bwd")
        return d_betas, d_rot.view(Fr, 24, 3, 3), None


class FusedLossFn(torch.autograd.Function):
    """lib/core/loss.py LossVideo/LossImage value AND gradient in two launches (maed_loss_fwd_bwd).
    Returns (total, losses[8]); only `total` is differentiable (the per-term entries are for logging, as in
    lib/core/trainer.py:209-211)."""

    @staticmethod
    def forward(ctx, pred_kp2d, pred_kp3d, pred_theta, gt_kp2d, gt_kp3d, gt_theta, w_smpl, skip, weights):
        """pred_* are (M2, ...) flattened frames; the last M3 = M2 - skip frames carry 3D / SMPL labels"""
        lib = L.lib()
        pred_kp2d, pred_kp3d, pred_theta = _f32(pred_kp2d), _f32(pred_kp3d), _f32(pred_theta)
        gt_kp2d, gt_theta = _f32(gt_kp2d), _f32(gt_theta)
        gt_kp3d = None if gt_kp3d is None else _f32(gt_kp3d)
        w8 = w_smpl.to(torch.uint8).contiguous()
        M2 = pred_kp2d.shape[0]
        M3 = M2 - skip
        dev = pred_kp2d.device
        alloc = torch.zeros if skip > 0 else torch.empty
        d_kp2d = torch.empty_like(pred_kp2d)
        d_kp3d = alloc(pred_kp3d.shape, dtype=torch.float32, device=dev)
        d_theta = alloc(pred_theta.shape, dtype=torch.float32, device=dev)
        losses = torch.empty(8, dtype=torch.float32, device=dev)
        partials = torch.empty(max(M2, 1) * 8, dtype=torch.float64, device=dev)
        lw = L.LossWeights(*[float(x) for x in weights])
        L.check(lib.maed_loss_fwd_bwd(ops._p(pred_kp2d), ops._p(gt_kp2d), M2, _off(pred_kp3d, skip * 147), ops._p(gt_kp3d),
                                      _off(pred_theta, skip * 85), ops._p(gt_theta), ops._p(w8), M3, C.byref(lw), ops._p(losses),
                                      ops._p(d_kp2d), _off(d_kp3d, skip * 147), _off(d_theta, skip * 85), ops._p(partials), ops._stream()),
                "loss_fwd_bwd")
        ctx.save_for_backward(d_kp2d, d_kp3d, d_theta)
        ctx.mark_non_differentiable(losses)
        return losses[5].clone(), losses

    @staticmethod
    def backward(ctx, g_total, _g_losses):
        d_kp2d, d_kp3d, d_theta = ctx.saved_tensors
        return d_kp2d * g_total, d_kp3d * g_total, d_theta * g_total, None, None, None, None, None, None


class AcclLossFn(torch.autograd.Function):
    """acceleration term of LossVideo (lib/core/loss.py:94-117) value and gradient in one launch (maed_loss_accl_fwd_bwd)"""

    @staticmethod
    def forward(ctx, pred_kp3d, gt_kp3d, weight):
        pred_kp3d, gt_kp3d = _f32(pred_kp3d), _f32(gt_kp3d)
        N, T = pred_kp3d.shape[:2]
        loss = torch.empty(1, dtype=torch.float64, device=pred_kp3d.device)
        d = torch.empty_like(pred_kp3d)
        L.check(L.lib().maed_loss_accl_fwd_bwd(ops._p(pred_kp3d), ops._p(gt_kp3d), N, T, float(weight), ops._p(loss), ops._p(d), ops._stream()), "loss_accl_fwd_bwd")
        ctx.save_for_backward(d)
        return loss[0].float()

    @staticmethod
    def backward(ctx, g):
        d, = ctx.saved_tensors
        return d * g, None, None
