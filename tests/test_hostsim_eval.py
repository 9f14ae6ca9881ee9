"""Evaluation metrics + Evaluator (maed_amd/eval_utils.py, maed_amd/evaluate.py, csrc/eval_metrics.hip; reference
lib/utils/eval_utils.py, lib/core/evaluate.py) against the fixture the reference's own functions and Evaluator produced
(tests/golden/g14_eval.npz, oracle/make_golden_eval.py).  Kernels run on the host simulator."""
import numpy as np
import pytest
import torch

from maed_amd import eval_utils as EU
from maed_amd.evaluate import Evaluator
from maed_amd.smpl import SMPL
from oracle import maed_ref as R
from oracle.make_golden_eval import StubModel

from _hostsim import patched


def t(a):
    return torch.from_numpy(np.asarray(a))


def close(got, want, rtol=1e-5, atol=1e-6):
    np.testing.assert_allclose(got.detach().cpu().numpy() if torch.is_tensor(got) else got, np.asarray(want), rtol=rtol, atol=atol)


def test_similarity_transform_matches_reference(golden):
    fx = golden("g14_eval")
    with patched():
        hat = EU.batch_compute_similarity_transform_torch(t(fx["S1"]), t(fx["S2"]))
        hat_t = EU.batch_compute_similarity_transform_torch(t(fx["S1"]).permute(0, 2, 1), t(fx["S2"]).permute(0, 2, 1))
    assert hat_t.shape == (12, 3, 14)
    close(hat_t.permute(0, 2, 1), hat, rtol=0, atol=0)
    # sets 4 and 5 are planar: rank-2 covariance, the fp32 SVD of the reference is itself only ~1e-4 accurate there
    close(hat, fx["S1_hat"], rtol=1e-4, atol=2e-4)
    close(hat[1], fx["S2"][1], atol=1e-5)                     # exact similarity copy is recovered exactly
    # mirrored set: a proper rotation cannot undo the reflection -> the aligned set must NOT coincide with the target
    assert float((hat[2] - t(fx["S2"][2])).abs().max()) > 0.1


def test_similarity_transform_is_a_proper_similarity():
    """size-independent property: S1_hat = s R S1 + t with R orthogonal, det +1, and it never does worse than the identity"""
    g = torch.Generator().manual_seed(0)
    S1, S2 = torch.randn(16, 17, 3, generator=g), torch.randn(16, 17, 3, generator=g)
    with patched():
        hat = EU.batch_compute_similarity_transform_torch(S1, S2)
    X, Y = S1 - S1.mean(1, keepdim=True), hat - hat.mean(1, keepdim=True)
    A = torch.linalg.lstsq(X, Y).solution                     # (16,3,3): Y = X A, A = s R^T
    s = torch.linalg.det(A).abs().pow(1 / 3)
    Rm = A / s[:, None, None]
    assert torch.allclose(Rm @ Rm.transpose(1, 2), torch.eye(3).expand(16, 3, 3), atol=1e-4)
    assert bool((torch.linalg.det(A) > 0).all())
    assert bool(((hat - S2).pow(2).sum((1, 2)) <= (S1 - S2).pow(2).sum((1, 2)) + 1e-5).all())


def test_accel_and_vertex_errors_match_reference(golden):
    fx = golden("g14_eval")
    with patched():
        close(EU.compute_accel(t(fx["acc_pred"])), fx["accel"])
        close(EU.compute_error_accel(t(fx["acc_gt"]), t(fx["acc_pred"])), fx["accel_err"])
        close(EU.compute_error_accel(t(fx["acc_gt"]), t(fx["acc_pred"]), vis=t(fx["acc_vis"])), fx["accel_err_vis"])
        close(EU.compute_error_verts(pred_verts=t(fx["verts_a"].astype(np.float32)), target_verts=t(fx["verts_b"].astype(np.float32))), fx["verts_err"])
        assert EU.compute_accel(torch.randn(2, 14, 3)).numel() == 0


def test_host_tensor_without_library_device_is_rejected():
    with pytest.raises(RuntimeError):
        try:
            EU.compute_accel(torch.randn(5, 14, 3))          # no GPU tensor, no simulator: must not hand host pointers to the GPU library
        except OSError as e:                                  # (library not built on this box: equally loud)
            raise RuntimeError(str(e))


def test_merge_and_interpolate_match_reference(golden):
    fx = golden("g14_eval")
    ev = Evaluator()
    merged = ev.merge_sequence([t(fx["merge_in0"]), t(fx["merge_in1"])])
    close(merged, fx["merge_out"], rtol=0, atol=0)
    close(ev.interpolate(merged, 11, 6), fx["interp_out"], rtol=1e-6, atol=1e-7)
    assert ev.interpolate(merged, 6, 6) is merged


def make_batch(fx):
    b = {k[len("batch."):]: fx[k] for k in fx.files if k.startswith("batch.")}
    out = {k: t(v) for k, v in b.items() if v.dtype.kind != "U"}
    out["instance_id"] = [list(r) for r in b["instance_id"]]
    out["paths"] = [list(r) for r in b["paths"]]
    return out


def test_evaluator_inference_and_evaluate_match_reference(golden):
    fx = golden("g14_eval")
    sp = R.make_synthetic_smpl(int(fx["smpl_seed"]))

    class DS:
        dataset_name = "mpii3d"

    class Loader(list):
        dataset = DS()

    model = StubModel(sp)
    model.decoder = torch.nn.Module()
    model.decoder.smpl = SMPL()                               # product SMPL on the same synthetic arrays (seed 0)
    ev = Evaluator()
    with patched():
        ev.inference(model, Loader([make_batch(fx)]), seqlen=3, interp=2, device="cpu", verbose=False)
        acc = {k: (torch.cat(v) if torch.is_tensor(v[0]) else np.concatenate(v)) for k, v in ev.evaluation_accumulators.items()}
        for k in ("pred_j3d", "pred_j2d", "pred_theta", "pred_rotmat", "target_j3d", "target_j2d", "target_theta", "bboxes"):
            close(acc[k], fx["acc." + k], rtol=1e-5, atol=1e-5)
        close(acc["pred_verts"][:, ::53], fx["acc.pred_verts_sub"], rtol=1e-5, atol=1e-5)
        assert list(acc["instance_id"]) == list(fx["acc.instance_id"]) and list(acc["paths"]) == list(fx["acc.paths"])
        eval_dict, num_pred = ev.evaluate()
    assert num_pred == int(fx["num_pred"])
    assert list(eval_dict) == ["mpjpe", "pa-mpjpe", "pve", "accel", "accel_err"]
    for k, v in eval_dict.items():
        assert abs(v - float(fx["eval." + k])) <= 2e-4 * abs(float(fx["eval." + k])), (k, v, float(fx["eval." + k]))
