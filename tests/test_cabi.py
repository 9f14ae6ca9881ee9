"""CPU checks of the drop-in boundary: libmaed_hip.so loads and exports every symbol include/maed_hip.h
declares, and the ctypes signature table mirrors the header (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "maed_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = re.findall(r"\b(?:int|size_t|const char\*)\s+(maed_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S)
    return {name: [a.strip() for a in args.split(",") if a.strip() and a.strip() != "void"] for name, args in decls}


@pytest.fixture(scope="module")
def built_lib():
    from maed_amd import build
    return build.build(verbose=False)


def test_header_declares_the_survey_minimum_set():
    names = set(declared_functions())
    for need in ["maed_layernorm_fwd", "maed_layernorm_bwd", "maed_gemm_nt", "maed_attn_spatial_fwd", "maed_attn_spatial_bwd",
                 "maed_attn_temporal_fwd", "maed_attn_temporal_bwd", "maed_st_mix_fwd", "maed_st_mix_bwd_apply", "maed_embed_add_fwd",
                 "maed_embed_add_bwd", "maed_ste_block_fwd", "maed_ste_block_bwd", "maed_ktd_chain_fwd", "maed_rot6d_pose_fwd",
                 "maed_smpl_lbs_fwd", "maed_joint_regress_fwd", "maed_smpl_joints_project_fwd", "maed_adam_step", "maed_last_error",
                 # SURVEY 8(b) backward / comm entries and the 8(f) rank-1 loss
                 "maed_ktd_chain_bwd", "maed_rot6d_pose_bwd", "maed_smpl_skin_bwd", "maed_smpl_chain_bwd", "maed_smpl_joints_project_bwd",
                 "maed_gemm_tn_wgrad", "maed_groupnorm_fwd", "maed_groupnorm_bwd", "maed_weight_std_fwd", "maed_weight_std_bwd",
                 "maed_loss_fwd_bwd", "maed_comm_init", "maed_comm_allreduce_async", "maed_comm_wait", "maed_comm_destroy"]:
        assert need in names, need


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/maed_hip.h but not exported"


def test_ctypes_table_matches_header(built_lib):
    from maed_amd import _lib
    decl = declared_functions()
    assert set(_lib.SIGNATURES) == set(decl), set(_lib.SIGNATURES) ^ set(decl)
    for name, args in decl.items():
        assert len(_lib.SIGNATURES[name][1]) == len(args), f"{name}: {len(_lib.SIGNATURES[name][1])} ctypes args vs {len(args)} declared"
    handle = _lib.lib()
    assert handle.maed_version() >= 100
    assert _lib.loaded_path() == built_lib


def test_no_cpu_fallback():
    """a CPU tensor must be an error, never a silent PyTorch path"""
    import torch
    from maed_amd import ops, _lib
    with pytest.raises(_lib.MaedHipError):
        ops.layernorm_fwd(torch.zeros(4, 128), torch.ones(128), torch.zeros(128), torch.float32)
