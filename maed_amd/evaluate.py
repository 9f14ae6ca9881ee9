"""Sliding-window inference and metrics (reference: lib/core/evaluate.py), same class / method names and the same
`evaluation_accumulators` / `eval_dict` keys -- SURVEY.md 8(f) rank 4.

Differences in mechanism, not in results: predictions stay on the GPU (the reference copies every tensor to numpy per
batch, evaluate.py:76-80), sub-clip merging and interpolation are torch ops, and MPJPE / PA-MPJPE / PVE / acceleration are
libmaed_hip kernels (maed_amd/eval_utils.py).  Joint regressors come from `data_dir` (.npy files, reference DATA_DIR) or from
the `j_regressors` dict; target vertices for PVE come from the model's own SMPL module.
"""
import os.path as osp
import time
from collections import defaultdict

import numpy as np
import torch

from . import eval_utils
from .smpl import H36M_TO_J14, H36M_TO_J17

# lib/models/smpl.py:65-81
J49_TO_MPII3D = list(range(25, 39)) + [39, 41, 43]
REGRESSOR_DICT = {'3dpw': 'J_regressor_h36m.npy', 'mpii3d': None, 'h36m': 'J_regressor_h36m.npy'}
JID_DICT = {'3dpw': H36M_TO_J14, 'h36m': H36M_TO_J17, 'mpii3d': J49_TO_MPII3D}


def move_dict_to_device(dic, device, tensor2float=False):
    """lib/utils/utils.py:21-29"""
    for k, v in dic.items():
        if isinstance(v, torch.Tensor):
            dic[k] = v.float().to(device) if tensor2float else v.to(device)
        elif isinstance(v, dict):
            move_dict_to_device(v, device)


def _flatten_dim(x):
    return x.reshape((-1,) + tuple(x.shape[2:]))


class Evaluator():
    def __init__(self, data_dir='data/smpl_data', j_regressors=None):
        self.evaluation_accumulators = defaultdict(list)
        self.data_dir = data_dir
        self.j_regressors = dict(j_regressors or {})
        self.smpl = None

    def _j_regressor(self, dataset_name):
        fname = REGRESSOR_DICT[dataset_name]
        if fname is None:
            return None
        if dataset_name in self.j_regressors:
            return torch.as_tensor(self.j_regressors[dataset_name]).float()
        return torch.from_numpy(np.load(osp.join(self.data_dir, fname))).float()

    def inference(self, model, dataloader, seqlen=8, interp=1, device='cuda', verbose=True, desc='[Evaluating] '):
        """evaluate.py:29-125.  interp (int >= 1): 1 out of <interp> frames is predicted by the model, the rest interpolated."""
        model.eval()
        dataset_name = dataloader.dataset.dataset_name
        start = time.time()
        self.evaluation_accumulators = defaultdict(list)
        self.smpl = getattr(getattr(model, 'decoder', None), 'smpl', self.smpl)
        J_regressor = self._j_regressor(dataset_name)
        Jid = JID_DICT[dataset_name]
        for target in dataloader:
            move_dict_to_device(target, device)
            with torch.no_grad():
                seqs = defaultdict(list)
                valid_joints = [j for j in range(target['kp_3d'].shape[2]) if target['kp_3d'][0, 0, j, -1]]
                orig_len = target['images'].shape[1]
                sub = target['images'][:, ::interp]
                interp_len = sub.shape[1]
                sample_freq = interp_len // seqlen
                for i in range(sample_freq):                      # strided sub-clips of seqlen frames (:68-71)
                    preds = model(sub[:, i::sample_freq], J_regressor=J_regressor)
                    seqs['verts'].append(preds['verts'])
                    seqs['j3d'].append(preds['kp_3d'][:, :, Jid])
                    seqs['j2d'].append(preds['kp_2d'][:, :, Jid])
                    seqs['theta'].append(preds['theta'])
                    seqs['rotmat'].append(preds['rotmat'])
                valid_seq = _flatten_dim(target['valid']).to(torch.bool)      # filters repeated (padding) frames
                for k in ('verts', 'theta', 'rotmat', 'j3d', 'j2d'):
                    merged = self.interpolate(self.merge_sequence(seqs[k]), orig_len, interp_len)
                    self.evaluation_accumulators['pred_' + k].append(merged[valid_seq.to(merged.device)])
                self.evaluation_accumulators['target_j3d'].append(_flatten_dim(target['kp_3d'][:, :, valid_joints])[valid_seq])
                self.evaluation_accumulators['target_j2d'].append(_flatten_dim(target['kp_2d'][:, :, valid_joints])[valid_seq])
                self.evaluation_accumulators['target_theta'].append(_flatten_dim(target['theta'])[valid_seq])
                vs = valid_seq.cpu().numpy()
                self.evaluation_accumulators['instance_id'].append(np.reshape(np.array(target['instance_id']).T, (-1,))[vs])
                self.evaluation_accumulators['paths'].append(np.reshape(np.array(target['paths']).T, (-1,))[vs])
                self.evaluation_accumulators['bboxes'].append(target['bbox'].reshape(-1, 4)[valid_seq])
        if verbose:
            print(f'{desc} | {time.time() - start:.2f}s')

    def merge_sequence(self, seq):
        """evaluate.py:127-133: interleave the strided sub-clips back into frame order, (N, T/k, k, ...) -> (N*T, ...)"""
        if seq is None:
            return None
        seq = torch.stack(list(seq), dim=2)
        assert seq.dim() >= 3
        return seq.reshape((-1,) + tuple(seq.shape[3:]))

    def interpolate(self, sequence, orig_len, interp_len):
        """evaluate.py:223-243: linear interpolation (with linear extrapolation at the ends, scipy interp1d
        fill_value='extrapolate') from interp_len samples at x = 1/L..1 to orig_len samples at x = 0..1; fp64 like scipy."""
        if orig_len == interp_len:
            return sequence
        L_ = interp_len
        seq = sequence.reshape((-1, L_) + tuple(sequence.shape[1:])).double()
        pos = torch.linspace(0., 1., orig_len, dtype=torch.float64, device=seq.device) * L_ - 1.0      # fractional sample index
        i0 = pos.floor().clamp(0, L_ - 2).long()
        w = (pos - i0).reshape((1, -1) + (1,) * (seq.dim() - 2))
        ret = seq[:, i0] * (1.0 - w) + seq[:, i0 + 1] * w
        return ret.reshape((-1,) + tuple(ret.shape[2:]))

    def evaluate(self, save_path=''):
        """evaluate.py:135-179 -> ({'mpjpe','pa-mpjpe','pve','accel','accel_err'} in mm, number of poses)"""
        acc = self.evaluation_accumulators
        for k, v in acc.items():
            acc[k] = torch.cat(v, dim=0) if torch.is_tensor(v[0]) else np.concatenate(v, axis=0)
        pred_j3ds, target_j3ds = acc['pred_j3d'], acc['target_j3d']
        num_pred = len(pred_j3ds)
        errors, errors_pa, pred_c, target_c = eval_utils.pose_errors(pred_j3ds, target_j3ds)
        pve = eval_utils.compute_error_verts(target_theta=acc['target_theta'], pred_verts=acc['pred_verts'], smpl=self.smpl)
        accel_err = eval_utils.compute_error_accel(joints_pred=pred_c, joints_gt=target_c)
        accel = eval_utils.compute_accel(pred_c)
        m2mm = 1000
        eval_dict = {
            'mpjpe': float(errors.mean()) * m2mm,
            'pa-mpjpe': float(errors_pa.mean()) * m2mm,
            'pve': float(pve.mean()) * m2mm,
            'accel': float(accel.mean()) * m2mm,
            'accel_err': float(accel_err.mean()) * m2mm,
        }
        if save_path:
            self.save_result(save_path, mpjpe=errors.cpu().numpy(), pa_mpjpe=errors_pa.cpu().numpy(), accel=accel_err.cpu().numpy())
        return eval_dict, num_pred

    def log(self, eval_dict, num_pred, desc=''):
        print(f"Evaluated on {int(num_pred)} number of poses.")
        print(f'{desc}' + ' '.join([f'{k.upper()}: {v:.4f},' for k, v in eval_dict.items()]))

    def run(self, model, dataloader, seqlen=8, interp=1, device='cuda', save_path='', verbose=True, desc='[Evaluating]'):
        self.inference(model, dataloader, seqlen=seqlen, interp=interp, device=device, verbose=verbose, desc=desc)
        eval_dict, num_pred = self.evaluate(save_path)
        self.log(eval_dict, num_pred)
        return eval_dict, num_pred

    def save_result(self, save_path, *args, **kwargs):
        """evaluate.py:207-221: inference.pkl with pred_theta, pred_verts, paths, bboxes + the per-frame errors"""
        import joblib
        save_fields = ['pred_theta', 'pred_verts', 'paths', 'bboxes']
        save_dic = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in self.evaluation_accumulators.items() if k in save_fields}
        save_dic.update(kwargs)
        joblib.dump(save_dic, osp.join(save_path, 'inference.pkl'))
