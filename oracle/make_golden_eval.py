"""Generate tests/golden/g14_eval.npz by running the REFERENCE's evaluation code (lib/core/evaluate.py Evaluator,
lib/utils/eval_utils.py), shim-imported from /root/reference -- SURVEY.md 8(f) rank 4.  Build container only:

    python -m oracle.make_golden_eval

Contents: (a) function-level vectors for the similarity transform (incl. reflected, planar and noisy-copy point sets), the
acceleration metrics and the vertex error; (b) Evaluator.merge_sequence / interpolate on random arrays; (c) a whole
Evaluator.inference + evaluate() run on a synthetic 'mpii3d' batch with a closed-form stub model (stub_model below, duplicated in
tests/test_hostsim_eval.py) and the stub SMPL of oracle/ref_shims.py for the PVE target vertices.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import maed_ref, ref_shims  # noqa: E402
from oracle.make_golden import save  # noqa: E402


def stub_model_outputs(inp, sp):
    """deterministic 'model': SMPL parameters are smooth functions of each frame's pixel statistics"""
    N, T = inp.shape[:2]
    m = inp.reshape(N * T, -1)
    feat = torch.stack([m.mean(1), m.std(1), m[:, ::7].mean(1), m[:, 1::5].mean(1)], dim=1)          # (NT,4)
    basis = torch.linspace(-1.0, 1.0, 4 * 82).reshape(4, 82)
    par = torch.tanh(feat @ basis)                                                                     # (NT,82)
    aa, betas = 0.4 * par[:, :72], par[:, 72:82]
    cam = torch.stack([0.9 + 0.05 * par[:, 0], 0.1 * par[:, 1], 0.1 * par[:, 2]], dim=1)
    ang = torch.norm(aa.reshape(-1, 3) + 1e-8, dim=1, keepdim=True)
    ax = aa.reshape(-1, 3) / ang
    K = torch.zeros(ax.shape[0], 3, 3)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -ax[:, 2], ax[:, 1], ax[:, 2], -ax[:, 0], -ax[:, 1], ax[:, 0]
    rot = (torch.eye(3)[None] + torch.sin(ang)[:, :, None] * K + (1 - torch.cos(ang))[:, :, None] * (K @ K)).reshape(-1, 24, 3, 3)
    verts, posed = maed_ref.smpl_lbs(betas, rot, sp)
    kp3d = maed_ref.smpl_joints49(verts, posed, sp)
    kp2d = maed_ref.projection(kp3d, cam)
    theta = torch.cat([cam, aa, betas], dim=1)
    return dict(theta=theta.reshape(N, T, -1), verts=verts.reshape(N, T, -1, 3), kp_2d=kp2d.reshape(N, T, -1, 2),
                kp_3d=kp3d.reshape(N, T, -1, 3), rotmat=rot.reshape(N, T, -1, 3, 3))


class StubModel(torch.nn.Module):
    def __init__(self, sp):
        super().__init__()
        self.sp = sp

    def forward(self, inp, J_regressor=None):
        assert J_regressor is None
        return stub_model_outputs(inp.cpu(), self.sp)


def synthetic_batch(g, N=2, T=12):
    kp3d = torch.cat([0.4 * torch.randn(N, T, 49, 3, generator=g), torch.ones(N, T, 49, 1)], -1)
    kp3d[..., 3] = 0.0
    kp3d[:, :, list(range(25, 39)) + [39, 41, 43], 3] = 1.0         # mpii3d labels exactly the joints JID_DICT['mpii3d'] selects
    kp3d[0, 5, 27, 3] = 0.0                                          # ... with a few joints invisible in later frames (masked, :142-146)
    kp3d[1, 2, 41, 3] = 0.0
    kp2d = torch.cat([torch.randn(N, T, 49, 2, generator=g), torch.ones(N, T, 49, 1)], -1)
    valid = torch.ones(N, T, dtype=torch.bool)
    valid[1, -3:] = False                                           # padding frames of the last clip
    theta = torch.cat([torch.randn(N, T, 3, generator=g) * 0.1, torch.randn(N, T, 72, generator=g) * 0.3, torch.randn(N, T, 10, generator=g)], -1)
    return dict(images=torch.randn(N, T, 3, 4, 4, generator=g), kp_3d=kp3d, kp_2d=kp2d, theta=theta, valid=valid,
                instance_id=[[f"vid{n}" for n in range(N)] for _ in range(T)], paths=[[f"vid{n}/{t:04d}.jpg" for n in range(N)] for t in range(T)],
                bbox=torch.rand(N, T, 4, generator=g))


def main():
    sp = ref_shims.install(smpl_seed=0)
    import lib.utils.eval_utils as eu
    from lib.core.evaluate import Evaluator
    g = torch.Generator().manual_seed(5)
    fx = {}

    # (a) functions ---------------------------------------------------------------------------------------------
    S2 = torch.randn(12, 14, 3, generator=g)
    S1 = torch.randn(12, 14, 3, generator=g)
    Rz = torch.tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    S1[1] = 1.7 * S2[1] @ Rz.T + torch.tensor([0.3, -0.2, 0.5])                       # exact similarity copy -> error 0
    S1[2] = S2[2] * torch.tensor([1.0, 1.0, -1.0])                                     # mirrored: the optimum needs det(R)=+1 handling
    S1[3] = S2[3] @ Rz.T + 0.01 * torch.randn(14, 3, generator=g)
    S1[4, :, 2] = 0.0                                                                  # planar source set (rank-2 covariance)
    S2[5, :, 1] = 0.25                                                                 # planar target set
    fx.update(S1=S1, S2=S2, S1_hat=eu.batch_compute_similarity_transform_torch(S1.clone(), S2.clone()))
    jp, jg = torch.randn(9, 14, 3, generator=g).numpy(), torch.randn(9, 14, 3, generator=g).numpy()
    vis = np.array([1, 1, 1, 0, 1, 1, 1, 1, 1], dtype=bool)
    fx.update(acc_pred=jp, acc_gt=jg, acc_vis=vis, accel=eu.compute_accel(jp), accel_err=eu.compute_error_accel(jg, jp),
              accel_err_vis=eu.compute_error_accel(jg, jp, vis))
    # vertex sets are stored as float16 (fixture size); the expected errors are computed on exactly those rounded values
    va, vb = (torch.randn(3, 6890, 3, generator=g).numpy().astype(np.float16) for _ in range(2))
    fx.update(verts_a=va, verts_b=vb, verts_err=eu.compute_error_verts(pred_verts=va.astype(np.float32), target_verts=vb.astype(np.float32)))

    # (b) Evaluator host logic ------------------------------------------------------------------------------------
    ev = Evaluator()
    parts = [torch.randn(2, 3, 5, 2, generator=g).numpy() for _ in range(2)]           # 2 strided sub-clips of 3 frames
    merged = ev.merge_sequence(parts)
    fx.update(merge_in0=parts[0], merge_in1=parts[1], merge_out=merged, interp_out=ev.interpolate(merged, 11, 6),
              interp_same=ev.interpolate(merged, 6, 6))

    # (c) inference + evaluate on a synthetic mpii3d batch ----------------------------------------------------------
    class DS:
        dataset_name = "mpii3d"

    class Loader(list):
        dataset = DS()

    batch = synthetic_batch(g)
    fx.update({"batch." + k: (v.numpy() if torch.is_tensor(v) else np.array(v)) for k, v in batch.items()})
    ev.inference(StubModel(sp), Loader([batch]), seqlen=3, interp=2, device="cpu", verbose=False)
    for k in ("pred_j3d", "pred_j2d", "pred_theta", "target_j3d", "target_j2d", "target_theta", "instance_id", "paths", "bboxes"):
        fx["acc." + k] = np.concatenate(ev.evaluation_accumulators[k], axis=0)
    fx["acc.pred_verts_sub"] = np.concatenate(ev.evaluation_accumulators["pred_verts"], axis=0)[:, ::53]
    fx["acc.pred_rotmat"] = np.concatenate(ev.evaluation_accumulators["pred_rotmat"], axis=0)
    eval_dict, num_pred = ev.evaluate()
    fx.update({"eval." + k: v for k, v in eval_dict.items()}, num_pred=num_pred, smpl_seed=0)
    save("g14_eval", **fx)


if __name__ == "__main__":
    main()
