"""Generate tests/golden/g8_rodrigues.npz and g11_loss.npz by running the REFERENCE's lib/core/loss.py and
lib/utils/geometry.py (shim-imported from /root/reference) on seeded inputs.  Build container only:

    python -m oracle.make_golden_loss

Stored: inputs, every entry of the reference's loss_dict, the total, and the gradients of the total w.r.t. the
three prediction tensors (reference autograd) -- so the GPU box needs no reference code.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_shims  # noqa: E402
from oracle.make_golden import save  # noqa: E402


def make_case(g, N2, N3, T, image=False):
    r = lambda *s: torch.randn(*s, generator=g)
    u = lambda *s: torch.rand(*s, generator=g)
    N = N2 + N3
    lead = (N,) if image else (N, T)
    lead3 = (N3,) if image else (N3, T)
    pshape = (N, 1) if image else (N, T)
    preds = dict(kp_2d=r(*pshape, 49, 2) * 0.5, kp_3d=r(*pshape, 49, 3) * 0.4,
                 theta=torch.cat([r(*pshape, 3) * 0.1 + 0.9, r(*pshape, 72) * 0.4, r(*pshape, 10)], -1))
    conf2 = (u(*lead3, 49, 1) > 0.3).float() * u(*lead3, 49, 1)
    conf3 = (u(*lead3, 49, 1) > 0.2).float()
    d3 = dict(kp_2d=torch.cat([r(*lead3, 49, 2) * 0.5, conf2], -1), kp_3d=torch.cat([r(*lead3, 49, 3) * 0.4, conf3], -1),
              theta=torch.cat([r(*lead3, 3), r(*lead3, 72) * 0.4, r(*lead3, 10)], -1), w_smpl=(u(*lead3) > 0.35).float())
    d2 = None
    if N2:
        d2 = dict(kp_2d=torch.cat([r(N2, T, 49, 2) * 0.5, u(N2, T, 49, 1)], -1))
    del lead
    return preds, d3, d2


def run(loss_mod, preds, *targets):
    leaves = {k: v.clone().requires_grad_(True) for k, v in preds.items()}
    total, d = loss_mod(leaves, *targets)
    total.backward()
    out = {"total": total.detach().numpy()}
    out.update({"term." + k: v.detach().numpy() for k, v in d.items()})
    out.update({"grad." + k: v.grad.numpy() for k, v in leaves.items()})
    out["term_order"] = np.array(list(d.keys()))
    return out


def main():
    ref_shims.install(smpl_seed=0)
    import lib.utils.geometry as geo
    from lib.core.loss import Loss, LossImage, LossVideo

    g = torch.Generator().manual_seed(11)
    aa = torch.randn(64, 3, generator=g)
    aa[0] = 0.0                       # the 1e-8 offset is what keeps this finite
    aa[1] = torch.tensor([3.1, 0.0, 0.0])
    aa[2] *= 1e-4
    aa[3] *= 4.0
    save("g8_rodrigues", axis_angle=aa.numpy(), rotmat=geo.batch_rodrigues(aa).numpy(),
         quat=torch.randn(16, 4, generator=torch.Generator().manual_seed(12)).numpy(),
         quat_rotmat=geo.quat2mat(torch.randn(16, 4, generator=torch.Generator().manual_seed(12))).numpy())

    fx = {}

    def put(prefix, preds, d3, d2, res):
        for k, v in preds.items():
            fx[f"{prefix}.pred.{k}"] = v.numpy()
        for k, v in d3.items():
            fx[f"{prefix}.d3.{k}"] = v.numpy()
        if d2:
            fx[f"{prefix}.d2.kp_2d"] = d2["kp_2d"].numpy()
        for k, v in res.items():
            fx[f"{prefix}.{k}"] = v

    # video, 2D-only clips in front of the 3D ones (trainer.py:253-262), default weights of Loss()
    preds, d3, d2 = make_case(g, 2, 3, 4)
    put("video_2d3d", preds, d3, d2, run(LossVideo(device="cpu"), preds, d3, d2))
    # video, 3D clips only
    preds, d3, d2 = make_case(g, 0, 3, 5)
    put("video_3d", preds, d3, None, run(LossVideo(device="cpu"), preds, d3, None))
    # video with the acceleration term switched on and other weights changed
    lv = LossVideo(e_loss_weight=5., e_3d_loss_weight=7., e_pose_loss_weight=2., e_shape_loss_weight=0.5, e_smpl_norm_loss=0.25,
                   e_smpl_accl_loss=3., device="cpu")
    put("video_accl", preds, d3, None, run(lv, preds, d3, None))
    # no frame with valid SMPL labels
    d3z = dict(d3, w_smpl=torch.zeros_like(d3["w_smpl"]))
    put("video_novalid", preds, d3z, None, run(LossVideo(device="cpu"), preds, d3z, None))
    # image loss (singleton T axis on the predictions), through the Loss front-end's weights
    preds, d3, _ = make_case(g, 0, 6, 1, image=True)
    front = Loss(device="cpu")
    put("image", preds, d3, None, run(front.loss_image, preds, d3))
    # Loss.forward dispatch + merge_loss
    pv, d3v, d2v = make_case(g, 1, 2, 3)
    lv_, dv_ = front(pv, target_3d=d3v, target_2d=d2v)
    li_, di_ = front(preds, target_img=d3)
    lm, dm = front.merge_loss(lv_, dv_, li_, di_, vid_w=0.7, img_w=1.3)
    for k, v in pv.items():
        fx[f"merge.pred.{k}"] = v.numpy()
    for k, v in d3v.items():
        fx[f"merge.d3.{k}"] = v.numpy()
    fx["merge.d2.kp_2d"] = d2v["kp_2d"].numpy()
    fx["merge.total"] = lm.detach().numpy()
    for k, v in dm.items():
        fx["merge.term." + k] = v.detach().numpy()
    # image loss on a mixed batch: some images without SMPL labels -- the reference's smpl_losses does NOT mask image input (loss.py:77)
    preds, d3, _ = make_case(g, 0, 6, 1, image=True)
    d3 = dict(d3, w_smpl=torch.tensor([1., 0., 1., 0., 1., 1.]))
    put("image_mixed", preds, d3, None, run(front.loss_image, preds, d3))
    save("g11_loss", **fx)


if __name__ == "__main__":
    main()
