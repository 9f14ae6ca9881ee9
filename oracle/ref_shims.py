"""Import shims that let the UNMODIFIED reference (/root/reference, torch-1.5 era) import under
torch 2.x in the BUILD CONTAINER ONLY.  Test infrastructure: used by oracle/make_golden.py and
oracle/time_reference_cpu.py to generate fixtures / CPU-baseline numbers.  Nothing on the GPU
box imports this file's targets (the reference does not travel).

Shims: torch._six, torchvision, yacs, smplx (absent third-party
packages), np.load of the absent J_regressor_extra.npy, and model_zoo.load_url.
The smplx stub's LBS is oracle.maed_ref.smpl_lbs on synthetic parameters ("parity unpinned").
"""
import collections
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


def install(smpl_seed=0):
    sys.dont_write_bytecode = True  # never drop __pycache__ into the read-only tree
    from oracle import maed_ref

    # (1) torch._six
    six = types.ModuleType("torch._six")
    six.container_abcs = collections.abc
    sys.modules["torch._six"] = six

    # (2) torchvision (only the 'cnn' encoder really needs it)
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvu = types.ModuleType("torchvision.models.utils")
    tvm.resnet50 = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("torchvision stub"))
    tvu.load_state_dict_from_url = lambda *a, **k: {}
    tv.models, tvm.utils = tvm, tvu
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.models.utils": tvu})

    # (3) yacs
    yacs = types.ModuleType("yacs")
    yc = types.ModuleType("yacs.config")

    class CfgNode(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

        def clone(self):
            return self

    yc.CfgNode = CfgNode
    yacs.config = yc
    sys.modules.update({"yacs": yacs, "yacs.config": yc})

    # (4) smplx
    sp = maed_ref.make_synthetic_smpl(smpl_seed)
    ModelOutput = collections.namedtuple(
        "ModelOutput", ["vertices", "joints", "full_pose", "betas", "global_orient", "body_pose"])
    ModelOutput.__new__.__defaults__ = (None,) * 6

    class SMPLStub(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            for name in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights"):
                self.register_buffer(name, sp[name].clone())
            self.faces = np.zeros((13776, 3), dtype=np.int64)

        def forward(self, betas=None, body_pose=None, global_orient=None, pose2rot=True, get_skin=True, **k):
            if pose2rot:    # smplx.lbs.batch_rodrigues on the (F,72) axis-angle pose: R = I + sin(a) K + (1 - cos(a)) K^2
                aa = torch.cat([global_orient, body_pose], dim=1).reshape(-1, 3)
                ang = torch.norm(aa + 1e-8, dim=1, keepdim=True)
                ax = aa / ang
                Kx = torch.zeros(aa.shape[0], 3, 3, dtype=aa.dtype)
                Kx[:, 0, 1], Kx[:, 0, 2], Kx[:, 1, 0] = -ax[:, 2], ax[:, 1], ax[:, 2]
                Kx[:, 1, 2], Kx[:, 2, 0], Kx[:, 2, 1] = -ax[:, 0], -ax[:, 1], ax[:, 0]
                s, c = torch.sin(ang)[:, :, None], torch.cos(ang)[:, :, None]
                rot = (torch.eye(3, dtype=aa.dtype)[None] + s * Kx + (1 - c) * (Kx @ Kx)).reshape(-1, 24, 3, 3)
                global_orient, body_pose = rot[:, :1], rot[:, 1:]
            rot = torch.cat([global_orient, body_pose], dim=1)
            verts, posed = maed_ref.smpl_lbs(betas, rot, sp)
            j45 = torch.cat([posed, verts[:, sp["extra_vertex_ids"]]], dim=1)
            return ModelOutput(vertices=verts, joints=j45, full_pose=rot, betas=betas,
                               global_orient=global_orient, body_pose=body_pose)

    smplx = types.ModuleType("smplx")
    smplx.SMPL = SMPLStub
    bm = types.ModuleType("smplx.body_models")
    bm.ModelOutput = ModelOutput
    lbs = types.ModuleType("smplx.lbs")
    lbs.vertices2joints = lambda J, v: torch.einsum("bik,ji->bjk", v, J)
    smplx.body_models, smplx.lbs = bm, lbs
    sys.modules.update({"smplx": smplx, "smplx.body_models": bm, "smplx.lbs": lbs})

    # (5) np.load of absent data files
    real_load = np.load

    def fake_load(path, *a, **k):
        if isinstance(path, str) and path.endswith("J_regressor_extra.npy"):
            return sp["J_regressor_extra"].numpy()
        return real_load(path, *a, **k)

    np.load = fake_load

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    # (6) pretrained download: the reference deletes head.* and loads with strict=False
    import lib.models.vision_transformer as vt
    vt.model_urls.setdefault("vit_base_resnet50_224_in21k", "none")
    vt.model_zoo.load_url = lambda *a, **k: {"head.weight": torch.zeros(1), "head.bias": torch.zeros(1)}
    return sp
