"""ctypes binding of libmaed_hip.so (C-ABI declared in include/maed_hip.h).

There is no fallback: if the library is missing the import of any op raises.  `lib()` loads it
lazily so that CPU-only tooling (state_dict manipulation, gloo tests of the host logic) can import
the package without a GPU; calling any kernel without the library is an error, never a silent
PyTorch path.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MAED_HIP_LIB") or os.path.join(HERE, "libmaed_hip.so")   # override: A/B builds of the same C-ABI

F32, BF16 = 0, 1
F32X3, F32X6, F32X1 = 2, 3, 4     # (F32X1: one bf16 plane -- backward products of the mixed mode) fp32 storage with an explicit matrix-product engine (split-bf16, 3 / 6 MFMAs per product): matrix-product entry points only
EPI_STORE, EPI_GELU, EPI_RESID_F32, EPI_MUL_DGELU, EPI_ATOMIC_F32, EPI_STORE_F32, EPI_TANH, EPI_ADD = range(8)
IMPL_AUTO, IMPL_VALU, IMPL_MFMA = 0, 1, 2
IMPL_MFMA_256 = 6       # gemm_nt only: 256x256 pipelined tiles
IMPL_MFMA_LONG = 5      # attention only: K/V-tiled long-sequence kernels
IMPL_MFMA_SK = 10      # gemm_nt only: persistent K-stream kernel (csrc/gemm_sk.hip)
IMPL_X3, IMPL_X6, IMPL_X1 = 7, 8, 9  # MAED_F32 matrix products on the bf16 matrix cores (split-bf16: 3 / 6 MFMAs per product), csrc/gemm_x3.hip
(OPT_F32_MATMUL, OPT_SIDE_STREAM, OPT_TN_TARGET_WGS, OPT_ABLATE, OPT_GN_BWD_ONEPASS, OPT_F32_BWD_X1, OPT_ST_FUSED, OPT_CONV3X3_ROWS_WGS, OPT_STEM_WGRAD_WGS,
 OPT_LBS_FRAMES, OPT_TN_DMA, OPT_X3_PLANES, OPT_X3_PLANES_LN, OPT_SK, OPT_SK_GRID, OPT_TN_SK, OPT_CONV3X3_NARROW_WGS, OPT_CONV3X3_FRAME) = range(18)   # maed_option (include/maed_hip.h)

vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float


class GnAffineItem(C.Structure):
    """maed_gn_affine_item (include/maed_hip.h): one layer of maed_gn_affine_grad_batch"""
    _fields_ = [("ab", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("N", C.c_int), ("C", C.c_int)]


class BlockDims(C.Structure):
    _fields_ = [("F", i32), ("P", i32), ("C", i32), ("H", i32), ("T", i32), ("hidden", i32),
                ("dtype", i32), ("impl", i32), ("eps", f32)]


_PARAM_FIELDS = ["ln1_g", "ln1_b", "ln2_g", "ln2_b", "w_qkv", "w_ts", "w_proj", "w_fc1", "w_fc2",
                 "b_qkv", "b_ts", "b_proj", "b_fc1", "b_fc2", "wt_qkv", "wt_ts", "wt_proj", "wt_fc1", "wt_fc2"]
_GRAD_FIELDS = _PARAM_FIELDS[:14]


class BlockParams(C.Structure):
    _fields_ = [(n, vp) for n in _PARAM_FIELDS]


class BlockGrads(C.Structure):
    _fields_ = [(n, vp) for n in _GRAD_FIELDS]


class SmplParams(C.Structure):
    _fields_ = [(n, vp) for n in ["v_template", "shapedirs", "posedirs", "J_template", "J_shapedirs", "lbs_weights", "parents"]]


class KtdPtrs(C.Structure):
    _fields_ = [("w", vp * 26), ("b", vp * 26), ("gw", vp * 26), ("gb", vp * 26)]


class LossWeights(C.Structure):
    _fields_ = [(n, f32) for n in ["w_kp2d", "w_kp3d", "w_pose", "w_shape", "w_norm"]]


KTD_W_ANC = 3420

# name -> (restype, argtypes): mirrors include/maed_hip.h one to one (tests/test_cabi.py checks it)
SIGNATURES = {
    "maed_last_error": (C.c_char_p, []),
    "maed_version": (i32, []),
    "maed_init": (i32, [i32]),
    "maed_device_faults": (i32, []),
    "maed_device_faults_clear": (i32, []),
    "maed_set_option": (i32, [i32, i32]),
    "maed_get_option": (i32, [i32]),
    "maed_layernorm_fwd": (i32, [vp, i64, vp, vp, vp, i32, vp, vp, i64, i32, f32, vp]),
    "maed_layernorm_bwd": (i32, [vp, i32, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp]),
    "maed_gemm_tn_wgrad": (i32, [vp, i64, vp, i64, i64, i32, i32, vp, i64, vp, i32, vp]),
    "maed_gemm_nt": (i32, [vp, i64, vp, i64, i64, i64, i64, i32, i32, vp, vp, i64, vp, vp, i64, i32, i32, vp]),
    "maed_transpose_cast": (i32, [vp, i32, i64, i64, i64, vp, i64, vp, i64, vp, i32, vp]),
    "maed_attn_spatial_fwd": (i32, [vp, vp, vp, i32, i32, i32, f32, i32, i32, vp]),
    "maed_attn_spatial_bwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, i32, vp]),
    "maed_attn_temporal_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, f32, i32, vp]),
    "maed_attn_temporal_bwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp]),
    "maed_st_colmean": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "maed_st_mix_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "maed_st_mix_bwd_reduce": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "maed_st_mix_bwd_apply": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "maed_st_fused_supported": (i32, [i32, i32, i32]),
    "maed_st_fused_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "maed_st_fused_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "maed_embed_add_fwd": (i32, [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "maed_embed_add_bwd": (i32, [vp, vp, i32, vp, vp, i32, i32, i32, i32, vp]),
    "maed_ste_block_saved_bytes": (C.c_size_t, [C.POINTER(BlockDims)]),
    "maed_ste_block_scratch_bytes": (C.c_size_t, [C.POINTER(BlockDims)]),
    "maed_gemm_nt_planes": (i32, [vp, vp, i64, vp, vp, i64, i64, i64, i64, i32, vp, vp, i64, vp, vp, i64, vp, vp, i32, vp]),
    "maed_split_planes": (i32, [vp, vp, vp, i64, vp]),
    "maed_ste_block_twin_work_bytes": (C.c_size_t, [C.POINTER(BlockDims)]),
    "maed_ste_block_fwd_twin": (i32, [C.POINTER(BlockDims), C.POINTER(BlockParams), vp, vp, vp, vp, i32, vp]),
    "maed_ste_block_twin_join": (i32, [vp]),
    "maed_ste_block_fwd": (i32, [C.POINTER(BlockDims), C.POINTER(BlockParams), vp, vp, vp, vp]),
    "maed_ste_block_infer": (i32, [C.POINTER(BlockDims), C.POINTER(BlockParams), vp, vp, vp, vp]),
    "maed_ste_block_bwd": (i32, [C.POINTER(BlockDims), C.POINTER(BlockParams), C.POINTER(BlockGrads), vp, vp, vp, vp, vp, vp, vp, vp]),
    "maed_loss_accl_fwd_bwd": (i32, [vp, vp, i32, i32, f32, vp, vp, vp]),
    "maed_dropout": (i32, [vp, vp, i64, f32, C.c_uint64, vp]),
    "maed_dropout_dev": (i32, [vp, vp, i64, f32, vp, C.c_uint64, vp]),
    "maed_tanh_bwd": (i32, [vp, vp, vp, i64, i32, vp]),
    "maed_stream_fence": (i32, [vp, vp]),
    "maed_prof_enable": (i32, [i32]),
    "maed_prof_ntags": (i32, []),
    "maed_prof_collect": (i32, [C.POINTER(C.c_double), C.POINTER(i32)]),
    "maed_prof_flops": (i32, [C.POINTER(C.c_double)]),
    "maed_prof_records": (i32, [i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), i32]),
    "maed_ktd_chain_fwd": (i32, [vp, vp, vp, i32, vp]),
    "maed_rot6d_pose_fwd": (i32, [vp, vp, vp, i64, vp]),
    "maed_smpl_lbs_fwd": (i32, [C.POINTER(SmplParams), vp, vp, vp, vp, vp, vp, i32, vp]),
    "maed_joint_regress_fwd": (i32, [vp, i32, vp, vp, i32, vp]),
    "maed_joint_regress_csr_fwd": (i32, [vp, vp, vp, i32, vp, vp, i32, vp]),
    "maed_smpl_joints_project_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp]),
    "maed_smpl_joints_project_bwd": (i32, [vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, i32, vp]),
    "maed_smpl_skin_bwd": (i32, [C.POINTER(SmplParams), vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "maed_smpl_skin_bwd_sparse": (i32, [C.POINTER(SmplParams), vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, vp]),
    "maed_smpl_chain_bwd": (i32, [C.POINTER(SmplParams), vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, i32, vp]),
    "maed_rot6d_pose_bwd": (i32, [vp, vp, vp, i64, vp, i64, vp]),
    "maed_ktd_chain_bwd": (i32, [vp, vp, vp, vp, vp, vp, i64, vp, vp, i32, vp]),
    "maed_ktd_pack": (i32, [C.POINTER(KtdPtrs), i32, vp, vp, vp, vp]),
    "maed_ktd_unpack_add": (i32, [C.POINTER(KtdPtrs), i32, vp, vp, vp, vp]),
    "maed_loss_fwd_bwd": (i32, [vp, vp, i32, vp, vp, vp, vp, vp, i32, C.POINTER(LossWeights), vp, vp, vp, vp, vp, vp]),
    "maed_weight_std_fwd": (i32, [vp, i32, i32, vp, i32, vp, f32, i32, vp]),
    "maed_weight_std_bwd": (i32, [vp, i32, i32, i32, vp, f32, vp]),
    "maed_groupnorm_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, i32, i32, vp]),
    "maed_groupnorm_fwd_twin": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, i32, vp, vp, vp]),
    "maed_groupnorm_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, i32, i32, vp, vp, vp]),
    "maed_gn_affine_grad_batch": (i32, [vp, i32, vp]),
    "maed_comm_load": (i32, [C.c_char_p]),
    "maed_comm_unique_id": (i32, [vp]),
    "maed_comm_init": (i32, [i32, i32, vp]),
    "maed_comm_allreduce_async": (i32, [vp, C.c_size_t, i32, vp]),
    "maed_comm_wait": (i32, [vp]),
    "maed_comm_world": (i32, []),
    "maed_comm_destroy": (i32, []),
    "maed_conv3x3_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, i32, vp, vp]),
    "maed_conv1x1_fwd": (i32, [vp, i64, vp, i64, i64, i32, i32, vp, i64, i32, vp, i32, vp]),
    "maed_conv3x3_tapmask": (i32, [vp, i32, i32, i32, vp]),
    "maed_conv3x3_s2_dgrad": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "maed_conv3x3_s2_tables": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "maed_conv3x3_s2_wgrad": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "maed_conv3x3_wgrad": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "maed_eval_pose_errors": (i32, [vp, vp, i32, i32, vp, vp, vp, vp, vp]),
    "maed_similarity_transform": (i32, [vp, vp, i32, i32, vp, vp]),
    "maed_eval_accel": (i32, [vp, vp, i32, i32, vp, vp]),
    "maed_eval_vertex_error": (i32, [vp, vp, i32, i32, vp, vp]),
    "maed_maxpool3s2_same_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "maed_maxpool3s2_same_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "maed_stem_input": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "maed_conv3x3_wgrad_rows64_scratch_floats": (i32, [i32, i32, i32, i32, i32]),
    "maed_conv3x3_wgrad_rows64": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "maed_stem7x7s2_supported": (i32, [i32, i32]),
    "maed_stem7x7s2_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "maed_stem7x7s2_wgrad_scratch_floats": (i32, [i32, i32, i32]),
    "maed_stem7x7s2_wgrad": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "maed_subsample2_fwd": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "maed_subsample2_bwd": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "maed_weight_refresh": (i32, [vp, i32, i32, i32, vp]),
    "maed_adam_step": (i32, [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, f32, f32, f32, vp]),
    "maed_adam_step_dev": (i32, [vp, vp, vp, vp, vp, i64, vp, f32, f32, f32, f32, f32, vp]),
}

_lib = None
_init_pending = False      # maed_init still to be called (lib(): as soon as the framework has a device context)


class MaedHipError(RuntimeError):
    pass


def lib():
    """Load libmaed_hip.so (once).  Raises if it has not been built: there is no fallback path."""
    global _lib, _init_pending
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MaedHipError(f"{LIB_PATH} is missing: build it with `python -m maed_amd.build` "
                               "(hipcc --offload-arch=gfx950).  maed_amd has no non-HIP fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
        apply_options(handle)
        _init_pending = True
    if _init_pending:
        # maed_init (architecture check: a GPU that is not gfx950 fails here, with its name, instead of at the first launch) needs a device context.  Binding the
        # library must not CREATE one: a multi-rank launcher that binds before torch.cuda.set_device(local_rank) would put a context of every rank on GPU 0
        #  -- so the check waits until the framework has initialised CUDA/HIP, and then looks at the device the caller selected.
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            _init_pending = False
            check(_lib.maed_init(torch.cuda.current_device()), "maed_init")
    return _lib


# what the host wants from the library's process-wide options (the library itself reads no environment variables).  The measurement
# knobs of README.md that live inside the library are translated here, once, when a library handle is bound.
_OPTIONS = {
    OPT_F32_MATMUL: {"exact": 0, "0": 0, "bf16x3": 1, "1": 1, "bf16x6": 2, "2": 2}[os.environ.get("MAED_F32_MATMUL", "exact")],
    OPT_SIDE_STREAM: int(os.environ.get("MAED_WGRAD_SIDE_STREAM", "1") == "1"),
    OPT_TN_TARGET_WGS: int(os.environ.get("MAED_TN_TARGET_WGS", "0")),
    OPT_ABLATE: int(os.environ.get("MAED_GEMM_ABLATE", "0")),
    OPT_F32_BWD_X1: int(os.environ.get("MAED_F32_BWD", "") == "bf16x1"),
    OPT_ST_FUSED: int(os.environ.get("MAED_ST_FUSED", "1")),              # A/B knob: 0 = the attentive addition as four launches per direction
    OPT_GN_BWD_ONEPASS: int(os.environ.get("MAED_GN_BWD_ONEPASS", "1")),      # A/B knob: 0 = the two-pass GroupNorm backward (2: 256-thread variant of the one-pass kernel)
    # row-item 3x3 weight gradient (64 -> 64 channels): MAED_CONV3X3_WGRAD_ROWS=0 sends the shape to the general TN kernel; ..._ROWS_WGS: workgroups per launch
    OPT_CONV3X3_ROWS_WGS: 0 if os.environ.get("MAED_CONV3X3_WGRAD_ROWS", "1") == "0" else int(os.environ.get("MAED_CONV3X3_ROWS_WGS", "256")),
    OPT_STEM_WGRAD_WGS: int(os.environ.get("MAED_STEM_WGS", "512")),
    OPT_LBS_FRAMES: int(os.environ.get("MAED_LBS_FB", "0")),
    OPT_TN_DMA: int(os.environ.get("MAED_TN_DMA", "1")),                 # A/B knob: 0 = the register-transposing weight-gradient kernel
    OPT_X3_PLANES: int(os.environ.get("MAED_X3_PLANES", "6")),           # twin forward of the STE block: fc1's activation stored as (hi, lo) planes, fc2 on maed_gemm_nt_planes' variant 2 / 4 / 5 / 6 / 7; 0 = fp32 activation + twin
    OPT_SK: int(os.environ.get("MAED_SK", "1")),                         # A/B knob: 0 = bf16 NT GEMMs on the per-tile kernels of rounds 1-5; 1 = persistent K-stream kernel by heuristic; 2 / 3 = forced
    OPT_SK_GRID: int(os.environ.get("MAED_SK_GRID", "0")),
    OPT_CONV3X3_FRAME: int(os.environ.get("MAED_CONV3X3_FRAME", "1")),              # 0: the stage-3 3x3 convolutions on 128 x 128 tiles (A/B knob)
    OPT_CONV3X3_NARROW_WGS: int(os.environ.get("MAED_CONV3X3_NARROW_WGS", "0")),   # 3x3 convolutions whose 128 x 128 grid is smaller than this take 128 x 64 tiles
    OPT_TN_SK: int(os.environ.get("MAED_TN_SK", "1")),                   # A/B knob: 0 = weight-gradient GEMMs on the split-M kernels with closing atomics (rounds 1-5)               # workgroups of the persistent kernel (0 = one per CU)
    OPT_X3_PLANES_LN: int(os.environ.get("MAED_X3_PLANES_LN", "0")),     # 1: ... and the two LayerNorm outputs as planes, qkv / fc1 on the plane kernel (measured neutral); 0 = fc2 only
}


def apply_options(handle):
    """push the host's option values into a freshly bound library (the product library, or the simulator the tests swap in)"""
    for k, v in _OPTIONS.items():
        if handle.maed_set_option(k, v) != 0:
            raise MaedHipError(f"maed_set_option({k}, {v}) failed")


def set_option(key, value):
    _OPTIONS[key] = int(value)
    if _lib is not None:
        check(_lib.maed_set_option(key, int(value)), "set_option")


def get_option(key):
    return _OPTIONS[key]


def check(rc, what=""):
    if rc != 0:
        msg = lib().maed_last_error().decode("utf-8", "replace")
        raise MaedHipError(f"{what or 'libmaed_hip'} failed with status {rc}: {msg}")


def device_faults():
    """frame-barrier timeouts the kernels reported since the process started (include/maed_hip.h maed_device_faults): 0 on a GPU this process has to itself.
    Non-zero: the affected step's result was NaN-poisoned and the library switched to its multi-launch forms; maed_last_error() says so."""
    return 0 if _lib is None else int(_lib.maed_device_faults())


def loaded_path():
    return LIB_PATH if _lib is not None else None
