"""Host-side operators: thin typed wrappers over the C-ABI (for tests and composition) and the
torch.autograd.Functions the nn.Modules in this package are made of.

Every function here enqueues HIP kernels on torch's CURRENT stream through libmaed_hip.so; there
is no PyTorch/CPU fallback (a CPU tensor or a missing library is an error).
"""
import ctypes as C
import math
import os
import threading
import weakref

import numpy as _np

import torch

from . import _lib as L
from ._lib import BF16, F32, check

# bumped by optimizers that update parameters through raw pointers (FusedAdam) so that cached
# compute-dtype / transposed weight copies are rebuilt
WEIGHT_EPOCH = 0


def bump_weight_epoch():
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


def on_library_device(t):
    """True for tensors the library can address: GPU tensors -- and host tensors while the test suite has swapped in the host simulator
    (tests/_hostsim.patched: the simulator library reports a negative maed_version()), so that module-level code paths that are
    otherwise GPU-only can be exercised without a GPU.  Never loads a library itself."""
    return t.is_cuda or (SIM_MODULE_PATHS and L._lib is not None and L._lib.maed_version() < 0)


SIM_MODULE_PATHS = True     # tests may switch the simulator's module-level paths off (tests/_hostsim.patched(module_paths=False))


# ---- fp32 matmul precision (the analogue of torch.set_float32_matmul_precision) -------------------------------------------------
# compute_dtype=torch.float32 keeps every activation and weight in fp32; what this chooses is the engine of the fp32 matrix products:
#   "exact"   fp32 FMA chains on the VALU (bit-for-bit parity mode; convolutions on MIOpen) -- the default
#   "bf16x3"  every operand split into two bf16 terms, three MFMAs per product, fp32 accumulation: error ~2^-16 per product (csrc/gemm_x3.hip)
#   "bf16x6"  three terms, six MFMAs: fp32-level error
# Process-wide (a library option: MAED_OPT_F32_MATMUL), like torch's flag; MAED_F32_MATMUL in the environment sets the initial value.
_F32_MODES = {"exact": 0, "highest": 0, "bf16x3": 1, "high": 1, "bf16x6": 2}


def set_float32_matmul_precision(mode):
    L.set_option(L.OPT_F32_MATMUL, _F32_MODES[mode])


def get_float32_matmul_precision():
    return ("exact", "bf16x3", "bf16x6")[L.get_option(L.OPT_F32_MATMUL)]


def f32_split():
    """True when fp32 matrix products run on the split-bf16 MFMA kernels: the library's own GEMM / convolution / weight-gradient entry points then
    take fp32 operands, and the f32 mode follows the bf16 mode's code paths (no transposed copies, no MIOpen)"""
    return L.get_option(L.OPT_F32_MATMUL) != 0


def lib_matmul_dtype(dtype, prec=None):
    """dtypes whose matrix products have library kernels for every role (forward, input gradient, weight gradient straight from row-major operands);
    prec: a module's own fp32 engine ("bf16x3" / "bf16x6", see mm_code), which counts like the process-wide mode"""
    return dtype == torch.bfloat16 or (dtype == torch.float32 and (f32_split() or prec in ("bf16x3", "bf16x6", "bf16x1")))


def bwd_prec(prec):
    """engine of the BACKWARD products (input gradient, weight gradient) of a module whose forward runs on "bf16x6": bf16x3.  What bf16x6 buys is the forward:
    rounding errors of the activations are re-normalised -- and amplified -- by every following GroupNorm, and the backward pass differentiates THOSE activations.
    The backward products themselves are linear in dy; their ~2^-16 error is not amplified (a weight gradient is even a leaf).  Measured on the full-size model
    against fp64 (scripts/x3_probe.py, profiles/r03_x3_probe_*.txt): backbone gradients with forward bf16x6 + backward bf16x3 sit exactly where all-bf16x6 does
    (median 1.60e-2 / 1.38e-2 / 2.7e-3 per stage = the fp32 oracle's own distance), all-bf16x3 at 7.0e-2 / 5.8e-2 / 7.9e-3; the step 50.8 -> 48.7 ms."""
    if L.get_option(L.OPT_F32_BWD_X1):
        return "bf16x1"
    return "bf16x3" if prec == "bf16x6" else prec


# "bf16": the backward of a compute_dtype = float32 model runs the bf16 MODE's backward on bf16 TWINS of what the forward saved (round 5):
# the forward keeps fp32 operands (split-bf16 products: the outputs' accuracy), the arena the backward reads holds bf16 copies written behind it.
_BWD_TWIN = [os.environ.get("MAED_F32_BWD", "") == "bf16"]


def set_float32_backward_precision(mode):
    """engine of the BACKWARD matrix products on fp32 tensors: None / "same" = the forward's (bf16x6 -> bf16x3, see bwd_prec), "bf16x1" = ONE bf16 plane per operand,
    one MFMA per product ("bf16x3 forward / bf16 backward", round 4): outputs keep the forward engine's accuracy (1e-3 on SMPL parameters and better), gradients are
    what the bf16 mode computes -- bf16 products, fp32 accumulation -- but from fp32-stored activations.  Process-wide (MAED_OPT_F32_BWD_X1: the fused STE block
    driver reads it); MAED_F32_BWD=bf16x1 in the environment sets the initial value.
    "bf16" (round 5): the backward is the bf16 MODE's, on bf16 twins of the saved activations (STE blocks: maed_ste_block_fwd_twin; backbone: the bf16 autograd graph
    over fp32 shadow tensors, maed_amd/resnetv2.py) -- bf16 operand traffic and the bf16 kernels instead of one-plane products on fp32 operands; whatever keeps fp32
    operands in its backward (the few GEMMs outside the blocks and the backbone) uses one plane."""
    assert mode in (None, "same", "bf16x1", "bf16"), mode
    L.set_option(L.OPT_F32_BWD_X1, int(mode in ("bf16x1", "bf16")))
    _BWD_TWIN[0] = mode == "bf16"


def get_float32_backward_precision():
    return "bf16" if _BWD_TWIN[0] else "bf16x1" if L.get_option(L.OPT_F32_BWD_X1) else "same"


# ---- fp32 shadows of a bf16 autograd graph (the backbone's half of the "bf16" backward mode) ---------------------------------------------------------------
# The backbone is a chain of per-layer autograd Functions; autograd insists that a gradient has the dtype of the tensor it belongs to.  For a bf16 backward behind an
# fp32 forward the GRAPH is therefore the bf16 mode's: every tensor autograd sees is the bf16 twin, and the fp32 tensor the forward chain really computes with travels
# beside it in this registry (data pointer of the twin -> (twin, fp32 shadow); the twin is held too, so its address cannot be recycled while the entry lives).  A
# Function's forward looks its inputs' shadows up, computes in fp32 (split-bf16 products), registers its output's shadow and SAVES THE TWINS; its backward is the bf16
# mode's, untouched.  The registry is emptied when the backbone's output has been consumed (HybridEmbed) -- the fp32 activations are transient.
_SHADOW = {}
TWIN_FORWARDS = [0]      # twin forwards taken (tests / bench.py assert that the mode they time is the mode that ran)


def shadow_put(t16, t32):
    _SHADOW[t16.data_ptr()] = (t16, t32)
    return t16


def shadow_of(t):
    if not _SHADOW or t is None or t.dtype != torch.bfloat16:
        return None
    e = _SHADOW.get(t.data_ptr())
    return e[1] if e is not None and e[0].numel() == t.numel() else None


def shadow_clear():
    assert not (_UNFILLED and _SHADOW) or all(k in _SHADOW for k in _UNFILLED), "an unfilled twin lost its shadow"
    for k in list(_UNFILLED):       # (twins nobody consumed: fill them before their shadows go)
        e = _SHADOW.get(k)
        if e is not None:
            e[0].copy_(e[1])
    _UNFILLED.clear()
    _SHADOW.clear()


def shadow_keep_only(t16):
    """drop every shadow but `t16`'s: the backbone calls it on its result -- the fp32 activations of its layers are dead once the last layer has run; only the output's
    shadow has a consumer left (the projection GEMM of HybridEmbed, which clears the registry)"""
    e = _SHADOW.get(t16.data_ptr()) if t16 is not None else None
    for k in list(_UNFILLED):
        if k in _SHADOW:
            _SHADOW[k][0].copy_(_SHADOW[k][1])
    _UNFILLED.clear()
    _SHADOW.clear()
    if e is not None:
        _SHADOW[t16.data_ptr()] = e


def twin_of(y32):
    """bf16 twin of an fp32 forward result, registered with its shadow (one cast pass)"""
    return shadow_put(y32.to(torch.bfloat16), y32)


_UNFILLED = set()    # data pointers of twins whose CONTENT the consumer will write (twin_later)


def twin_later(y32):
    """bf16 twin of a convolution's fp32 result WITHOUT a cast pass: every convolution of the backbone feeds a GroupNorm, whose forward reads the fp32 tensor anyway
    and writes the twin from its registers (maed_groupnorm_fwd_twin twin_x).  Until then the twin is an identity for autograd and the shadow registry only; a
    consumer that is not a GroupNorm (none in this backbone) fills it with a cast (twin_fill)."""
    y16 = torch.empty_like(y32, dtype=torch.bfloat16)
    _UNFILLED.add(y16.data_ptr())
    return shadow_put(y16, y32)


def twin_fill(t16):
    """make sure a twin_later twin holds data (consumers other than GroupNormFn)"""
    if t16 is not None and t16.data_ptr() in _UNFILLED:
        _UNFILLED.discard(t16.data_ptr())
        t16.copy_(shadow_of(t16))


def bwd_twin():
    """True: compute_dtype = float32 modules save bf16 twins and run the bf16 mode's backward (needs the split-bf16 forward engine)"""
    return _BWD_TWIN[0] and f32_split()


def mm_code(dtype, prec=None):
    """dtype code for the matrix-product entry points: fp32 tensors of a module with its own engine (ResNetV2.f32_matmul) go as MAED_F32X3 / MAED_F32X6
    (backward products of the mixed mode: MAED_F32X1), everything else as dt_code (fp32 then follows the process-wide mode)"""
    if dtype == torch.float32 and prec in ("bf16x3", "bf16x6", "bf16x1"):
        return {"bf16x3": L.F32X3, "bf16x6": L.F32X6, "bf16x1": L.F32X1}[prec]
    return dt_code(dtype)


def dt_code(dtype):
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"maed_amd: unsupported compute dtype {dtype}")


_TLS = threading.local()     # .stream: raw handle the wrappers launch on instead of the current stream (side_stream_run), per thread


def _stream():
    s = getattr(_TLS, "stream", None)
    if s is not None:
        return s
    # raw handle of the current stream of the current device: one C call (torch.cuda.current_stream() builds a Stream object and resolves the device index
    # through Python -- 11 us a piece, 130 of them per train step: scripts/host_profile.py)
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


# ---- side stream for the backbone's weight-gradient GEMMs -----------------------------------------------------------------------------
# dW += Y^T X of a convolution depends only on tensors that exist when its backward runs, and nothing reads dW before the batched
# weight-standardisation backward.  On the caller's stream the TN kernel (bound by its LDS-side pipeline) sits between
# the input-gradient GEMM and the GroupNorm backward of the layer in front (HBM-bound): on a second stream it runs beside them.
# The operands stay referenced until side_stream_join() has ordered the caller's stream after everything issued so far (_keep_until_join; WeightStdFn.backward
# joins before it reads the dW slices, and one join is queued for the end of every backward pass), so the caching allocator cannot hand their memory out early.
# MAED_WGRAD_SIDE_STREAM=0: everything on the caller's stream (A/B knob; the STE blocks' C++ driver reads the same variable).
_SIDE_ON = None      # None: follow the library option (read at USE time -- a later _lib.set_option(OPT_SIDE_STREAM) then moves the Python side with the C++ block driver);
                     # True / False: override (bench.py's single-stream profiling steps, A/B scripts, tests)
_SIDE = {}


def _side_on():
    return (L.get_option(L.OPT_SIDE_STREAM) == 1) if _SIDE_ON is None else _SIDE_ON


def _join_at_end_of_backward(device, st):
    """whoever reads the side stream's results without going through WeightStdFn.backward (GroupNorm parameters of a backbone whose convolutions are
    frozen, a caller's own optimizer reading .grad right after backward()) is covered by one join queued for the end of the running backward pass"""
    if st[1]:
        return                          # already pending: the callback of this pass is queued
    try:
        torch.autograd.Variable._execution_engine.queue_callback(lambda: side_stream_join(device))
    except RuntimeError:
        pass                            # not inside a backward pass (direct calls from tests / micro-benchmarks): the caller joins


def side_stream_run(fn, *tensors):
    t0 = tensors[0]
    if not (t0.is_cuda and _side_on()):
        return fn()
    st = _SIDE.get(t0.device)
    if st is None:
        st = _SIDE[t0.device] = [torch.cuda.Stream(device=t0.device), False]
    side = st[0]
    # fn() is library launches only: they take the side stream's raw handle from _stream() (thread-local override) after one fence behind the caller's stream --
    # no framework stream switch, no Python-level event (side.wait_stream + `with torch.cuda.stream(side)` cost 50 us per use, 49 uses per train step)
    raw = side.cuda_stream
    check(L.lib().maed_stream_fence(_stream(), raw), "stream_fence")
    prev = getattr(_TLS, "stream", None)
    _TLS.stream = raw
    try:
        out = fn()
    finally:
        _TLS.stream = prev
    _keep_until_join(t0.device, st, tensors)
    _join_at_end_of_backward(t0.device, st)
    st[1] = True
    return out


def side_stream_handle(device, *tensors):
    """raw handle of the side stream for entry points that fence a trailing kernel onto it themselves (maed_groupnorm_bwd aux_stream);
    None when the side stream is off.  `tensors`: what that kernel reads / writes (kept alive for it)."""
    if not (device.type == "cuda" and _side_on()):
        return None
    st = _SIDE.get(device)
    if st is None:
        st = _SIDE[device] = [torch.cuda.Stream(device=device), False]
    _keep_until_join(device, st, tensors)
    _join_at_end_of_backward(device, st)
    st[1] = True
    return st[0].cuda_stream


def _keep_until_join(device, st, tensors):
    """what a side-stream launch reads / writes stays referenced until the caller's stream has been ordered behind the side stream (side_stream_join): freed
    after that, the caching allocator hands the memory out for work that is enqueued behind the join.  Tensor.record_stream would do the same job per tensor, but
    defers the reuse until the side stream's EVENT has completed on the device: a host that runs several steps ahead of the GPU (13 ms of enqueue against a 22 ms
    step) then finds nothing reusable and the allocator keeps growing -- 10 -> 30 GB reserved over 12 free-running steps, with hipMalloc stalls of 100+ ms in the
    fp32 modes (scripts/alloc_probe.py)."""
    keep = _KEEP.setdefault(device, [])
    keep.extend(tensors)
    if len(keep) > 4096:                # nobody joined for a long time (direct calls outside a backward pass): join now
        side_stream_join(device)


_KEEP = {}


def _keep_alive_on_stream(t):
    """scratch a wrapper allocates for a launch that may be running on the side stream (side_stream_run's thread-local override): referenced until the join, like
    the operands (_keep_until_join); on the caller's own stream the caching allocator's stream order already covers it"""
    if getattr(_TLS, "stream", None) is not None:
        _KEEP.setdefault(t.device, []).append(t)


def side_stream_join(device):
    st = _SIDE.get(device)
    if st is not None and st[1]:
        torch.cuda.current_stream(device).wait_stream(st[0])
        st[1] = False
    keep = _KEEP.get(device)
    if keep:
        keep.clear()


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise L.MaedHipError("maed_amd kernels need CUDA/HIP tensors (no CPU path)")
    return t.data_ptr()


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


_TABLES = {}            # (device, table bytes) -> device copy: the pointer tables of a step are the same from step to step (parameters, gradients and arenas do not move)
_CAPTURE_KEEP = None    # list owned by a capturing graphed.GraphedTrainStep: what a captured launch / copy node reads must outlive the graph


def _upload_table(tab, dev):
    """small host-built pointer table -> device WITHOUT a host/device synchronisation point: staged in pinned memory and copied
    asynchronously on the current stream (a pageable .to(device) blocks the host until the stream has drained, so the GPU
    idled ~1 ms per step at three such uploads); the caching host allocator keeps the pinned block alive until the copy ran.
    Round 6: a table with the same bytes as an earlier one re-uses that one's device copy (immutable once uploaded) -- in steady state no step uploads anything."""
    host = torch.from_numpy(tab.view(_np.uint8))
    if dev.type != "cuda":
        return host.clone()
    key = (dev, host.numpy().tobytes())
    capturing = torch.cuda.is_current_stream_capturing()
    hit = _TABLES.get(key)
    if hit is not None:
        if capturing and _CAPTURE_KEEP is not None:
            _CAPTURE_KEEP.append(hit)
        return hit
    if capturing:
        # a host -> device copy inside a capture would have to come out of the framework's pinned-memory cache, whose bookkeeping (an event per block, queried at the
        # next allocation) does not survive events recorded in a capturing stream (hipErrorCapturedEvent): the tables of a replayable step must be the ones its eager
        # warm-up steps uploaded -- they are, as long as no table row points at a per-step temporary (WeightStdFn keeps its fp32 dW arena for this reason)
        raise RuntimeError("maed_amd: a pointer table changed between the eager warm-up steps and the capture (a row points at a per-step temporary)")
    pinned = host.pin_memory()
    out = pinned.to(dev, non_blocking=True)
    if len(_TABLES) >= 64:
        _TABLES.pop(next(iter(_TABLES)))        # oldest entry: its device copy is freed in stream order behind the launches that read it
    _TABLES[key] = out
    return out


# ----------------------------------------------------------------------------------------------
# primitive wrappers (no autograd)
# ----------------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, out_dtype, eps=1e-6, row_stride=None, rows=None):
    C_ = x.shape[-1]
    if rows is None:
        x = _c(x)
        rows = x.numel() // C_
        row_stride = C_
    y = torch.empty(rows, C_, dtype=out_dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    check(L.lib().maed_layernorm_fwd(_p(x), row_stride, _p(gamma), _p(beta), _p(y), dt_code(out_dtype), _p(mean), _p(rstd),
                                     rows, C_, eps, _stream()), "layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dres=None, dgamma=None, dbeta=None, want_twin=False):
    C_ = x.shape[-1]
    x, dy = _c(x), _c(dy)
    rows = x.numel() // C_
    dx = torch.empty(rows, C_, dtype=torch.float32, device=x.device)
    twin = torch.empty(rows, C_, dtype=dy.dtype, device=x.device) if want_twin else None
    dgamma = torch.zeros(C_, dtype=torch.float32, device=x.device) if dgamma is None else dgamma
    dbeta = torch.zeros(C_, dtype=torch.float32, device=x.device) if dbeta is None else dbeta
    check(L.lib().maed_layernorm_bwd(_p(dy), dt_code(dy.dtype), _p(x), C_, _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), _p(twin),
                                     _p(dgamma), _p(dbeta), rows, C_, _stream()), "layernorm_bwd")
    return (dx, dgamma, dbeta, twin) if want_twin else (dx, dgamma, dbeta)


def gemm_tn_wgrad(Y, X, dW=None, dbias=None, prec=None):
    """dW[N,K] += Y[M,N]^T X[M,K]; dbias[N] += colsum(Y)   (bf16 operands, or fp32 ones on the split-bf16 kernels; fp32 accumulators)"""
    M, N = Y.shape
    K = X.shape[1]
    dW = torch.zeros(N, K, dtype=torch.float32, device=Y.device) if dW is None else dW
    check(L.lib().maed_gemm_tn_wgrad(_p(Y), Y.stride(0), _p(X), X.stride(0), M, N, K, _p(dW), dW.stride(0), _p(dbias), mm_code(Y.dtype, prec), _stream()),
          "gemm_tn_wgrad")
    return dW


def gemm_nt(A, B, epilogue=L.EPI_STORE, bias=None, out=None, out2=None, aux=None, splitk=1, impl=L.IMPL_AUTO, M=None, K=None, prec=None):
    """out = epilogue(A[M,K] @ B[N,K]^T).  A/B in the compute dtype (f32 or bf16); bias fp32.
    EPI_ADD: out2 (optional, uint8) = 1 bit per element of aux: aux is masked by it before the add (GroupNormFn's lazily masked residual gradient)."""
    assert A.dim() == 2 and B.dim() == 2 and A.dtype == B.dtype
    M = A.shape[0] if M is None else M
    K = A.shape[1] if K is None else K
    N = B.shape[0]
    assert A.stride(1) == 1 and B.stride(1) == 1
    if out is None:
        odt = torch.float32 if epilogue in (L.EPI_RESID_F32, L.EPI_ATOMIC_F32, L.EPI_STORE_F32) else A.dtype
        out = (torch.zeros if epilogue == L.EPI_ATOMIC_F32 else torch.empty)(M, N, dtype=odt, device=A.device)
    if epilogue == L.EPI_GELU and out2 is None:
        out2 = torch.empty_like(out)
    check(L.lib().maed_gemm_nt(_p(A), A.stride(0), _p(B), B.stride(0), M, N, K, mm_code(A.dtype, prec), epilogue, _p(bias),
                               _p(out), out.stride(0), _p(out2), _p(aux), aux.stride(0) if aux is not None else 0,
                               splitk, impl, _stream()), "gemm_nt")
    return (out, out2) if epilogue == L.EPI_GELU else out


def split_planes(x):
    """(hi, lo) bf16 planes of an fp32 tensor: hi = bf16(x), lo = bf16(x - hi) -- the operand format of gemm_nt_planes (csrc/gemm_x3p.hip)"""
    x = _c(x)
    assert x.dtype == torch.float32 and x.numel() % 8 == 0
    hi, lo = torch.empty_like(x, dtype=torch.bfloat16), torch.empty_like(x, dtype=torch.bfloat16)
    check(L.lib().maed_split_planes(_p(x), _p(hi), _p(lo), x.numel(), _stream()), "split_planes")
    return hi, lo


def gemm_nt_planes(A, B, epilogue=L.EPI_STORE, bias=None, aux=None, want_f32=True, want_planes=False, want_pre=False, variant=0):
    """epilogue((A_hi + A_lo)[M,K] @ (B_hi + B_lo)[N,K]^T) in the bf16x3 arithmetic on plane operands; A, B = (hi, lo) pairs of 2-D bf16 tensors.
    Returns (out fp32 | None, (hi, lo) | None, pre-activation bf16 | None)."""
    (Ah, Al), (Bh, Bl) = A, B
    assert Ah.dtype == Al.dtype == Bh.dtype == Bl.dtype == torch.bfloat16 and Ah.shape == Al.shape and Bh.shape == Bl.shape
    assert Ah.stride() == Al.stride() and Bh.stride() == Bl.stride() and Ah.stride(1) == 1 and Bh.stride(1) == 1
    M, K = Ah.shape
    N = Bh.shape[0]
    dev = Ah.device
    out = torch.empty(M, N, dtype=torch.float32, device=dev) if want_f32 else None
    oh = torch.empty(M, N, dtype=torch.bfloat16, device=dev) if want_planes else None
    ol = torch.empty(M, N, dtype=torch.bfloat16, device=dev) if want_planes else None
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=dev) if (want_pre and epilogue == L.EPI_GELU) else None
    check(L.lib().maed_gemm_nt_planes(_p(Ah), _p(Al), Ah.stride(0), _p(Bh), _p(Bl), Bh.stride(0), M, N, K, epilogue, _p(bias), _p(out), N, _p(pre), _p(aux),
                                      aux.stride(0) if aux is not None else 0, _p(oh), _p(ol), variant, _stream()), "gemm_nt_planes")
    return out, ((oh, ol) if want_planes else None), pre


def transpose_cast(x, out_dtype, want_t=True, want_c=False, colsum=None, pad_to=64):
    """x (M,N) f32/bf16 -> (x^T (N, Mp) zero-padded, cast copy (M,N)) in out_dtype; colsum[N] += sum_m."""
    assert x.dim() == 2 and x.stride(1) == 1
    M, N = x.shape
    Mp = (M + pad_to - 1) // pad_to * pad_to
    out_t = torch.empty(N, Mp, dtype=out_dtype, device=x.device) if want_t else None
    out_c = torch.empty(M, N, dtype=out_dtype, device=x.device) if want_c else None
    check(L.lib().maed_transpose_cast(_p(x), dt_code(x.dtype), x.stride(0), M, N, _p(out_t), Mp, _p(out_c), N, _p(colsum),
                                      dt_code(out_dtype), _stream()), "transpose_cast")
    return out_t, out_c


def attn_spatial_fwd(qkv, H, impl=L.IMPL_AUTO):
    F_, P, C3 = qkv.shape
    C_ = C3 // 3
    qkv = _c(qkv)
    o = torch.empty(F_, P, C_, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(F_, H, P, dtype=torch.float32, device=qkv.device)
    check(L.lib().maed_attn_spatial_fwd(_p(qkv), _p(o), _p(lse), F_, P, H, 1.0 / math.sqrt(C_ // H), dt_code(qkv.dtype), impl, _stream()),
          "attn_spatial_fwd")
    return o, lse


def attn_spatial_bwd(qkv, o, d_o, lse, H, dqkv=None, accumulate=False, impl=L.IMPL_AUTO):
    F_, P, C3 = qkv.shape
    dqkv = torch.empty_like(qkv) if dqkv is None else dqkv
    check(L.lib().maed_attn_spatial_bwd(_p(_c(qkv)), _p(_c(o)), _p(_c(d_o)), _p(lse), _p(dqkv), int(accumulate), F_, P, H,
                                        1.0 / math.sqrt(C3 // 3 // H), dt_code(qkv.dtype), impl, _stream()), "attn_spatial_bwd")
    return dqkv


def attn_temporal_fwd(qkv, H, T, prec=None):
    """prec (fp32 only): "bf16x3" = split-bf16 contractions on the matrix cores for this call (one-tile sequences: 32 % T == 0), None = the process-wide mode"""
    F_, P, C3 = qkv.shape
    C_ = C3 // 3
    qkv = _c(qkv)
    o = torch.empty(F_, P, C_, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(F_, H, P, dtype=torch.float32, device=qkv.device)
    check(L.lib().maed_attn_temporal_fwd(_p(qkv), _p(o), _p(lse), F_, P, H, T, 1.0 / math.sqrt(C_ // H), mm_code(qkv.dtype, prec), _stream()),
          "attn_temporal_fwd")
    return o, lse


def attn_temporal_bwd(qkv, o, d_o, lse, H, T, dqkv=None, accumulate=False, prec=None):
    F_, P, C3 = qkv.shape
    dqkv = torch.empty_like(qkv) if dqkv is None else dqkv
    check(L.lib().maed_attn_temporal_bwd(_p(_c(qkv)), _p(_c(o)), _p(_c(d_o)), _p(lse), _p(dqkv), int(accumulate), F_, P, H, T,
                                         1.0 / math.sqrt(C3 // 3 // H), mm_code(qkv.dtype, prec), _stream()), "attn_temporal_bwd")
    return dqkv


def st_colmean(x_s, x_t):
    F_, P, C_ = x_s.shape
    means = torch.empty(F_, 2 * C_, dtype=x_s.dtype, device=x_s.device)
    ws = torch.empty(F_, 2 * C_, dtype=torch.float32, device=x_s.device)
    check(L.lib().maed_st_colmean(_p(_c(x_s)), _p(_c(x_t)), _p(means), _p(ws), F_, P, C_, dt_code(x_s.dtype), _stream()), "st_colmean")
    return means


def st_mix_fwd(x_s, x_t, logits):
    F_, P, C_ = x_s.shape
    mix = torch.empty_like(x_s)
    check(L.lib().maed_st_mix_fwd(_p(_c(x_s)), _p(_c(x_t)), _p(_c(logits)), _p(mix), F_, P, C_, dt_code(x_s.dtype), _stream()), "st_mix_fwd")
    return mix


def st_mix_bwd(dmix, x_s, x_t, logits, dmeans_fn):
    """dmeans_fn(dlogits) -> dmeans (both (F,2C) in the compute dtype): the ts_attn backward GEMM."""
    F_, P, C_ = x_s.shape
    dlogits = torch.empty(F_, 2 * C_, dtype=x_s.dtype, device=x_s.device)
    ws = torch.empty(F_, 2 * C_, dtype=torch.float32, device=x_s.device)
    check(L.lib().maed_st_mix_bwd_reduce(_p(_c(dmix)), _p(_c(x_s)), _p(_c(x_t)), _p(logits), _p(dlogits), _p(ws), F_, P, C_,
                                         dt_code(x_s.dtype), _stream()), "st_mix_bwd_reduce")
    dmeans = dmeans_fn(dlogits)
    dx_s, dx_t = torch.empty_like(x_s), torch.empty_like(x_t)
    check(L.lib().maed_st_mix_bwd_apply(_p(_c(dmix)), _p(logits), _p(_c(dmeans)), _p(dx_s), _p(dx_t), F_, P, C_, dt_code(x_s.dtype), _stream()),
          "st_mix_bwd_apply")
    return dx_s, dx_t, dlogits


def adam_step(p, g, m, v, shadow, lr, beta1, beta2, eps, wd, step, gscale=1.0):
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    check(L.lib().maed_adam_step(_p(p), _p(g), _p(m), _p(v), _p(shadow), p.numel(), lr, beta1, beta2, eps, wd, bc1, bc2, gscale, _stream()),
          "adam_step")


def adam_step_dev(p, g, m, v, shadow, state, beta1, beta2, eps, wd, gscale=1.0):
    """maed_adam_step with lr / bias corrections read from the device record `state` (DeviceTrainState.dev) when the kernel runs"""
    check(L.lib().maed_adam_step_dev(_p(p), _p(g), _p(m), _p(v), _p(shadow), p.numel(), _p(state), beta1, beta2, eps, wd, gscale, _stream()), "adam_step_dev")


class DeviceTrainState:
    """include/maed_hip.h `maed_train_state`: the per-step scalars of a training step (learning rate, Adam's bias corrections, the Dropout seed) in a 32-byte
    device record, so that the launches of a step carry no argument that changes from step to step -- the precondition of replaying the step as a hipGraph
    (maed_amd/graphed.py).  The host fills the mirror and uploads it (one 32-byte copy on the launch stream) before the step's kernels are enqueued / replayed."""

    def __init__(self, device):
        self.dev = torch.zeros(32, dtype=torch.uint8, device=device)
        self._host = _np.zeros(32, dtype=_np.uint8)
        self.calls = 0          # Dropout layers seen so far in the running step (call_id of maed_dropout_dev)

    def set_hyper(self, lr, bias_corr1, bias_corr2):
        self._host[:12].view(_np.float32)[:] = (lr, bias_corr1, bias_corr2)

    def begin_step(self, seed=None):
        """new Dropout seed (from torch's CPU generator unless given: torch.manual_seed reproduces a run), Dropout call counter back to zero"""
        self.calls = 0
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        self._host[16:24].view(_np.uint64)[0] = seed

    def upload(self):
        # a fresh pinned block per upload: the host runs steps ahead of the GPU, a re-used staging buffer would be overwritten before its copy has run
        src = torch.from_numpy(self._host.copy())
        self.dev.copy_(src.pin_memory() if self.dev.is_cuda else src, non_blocking=True)

    def next_call_id(self):
        self.calls += 1
        return self.calls


DEVICE_STATE = None      # DeviceTrainState of the running training loop (graphed.GraphedTrainStep sets it): Dropout then takes its seed from the device record


# ----------------------------------------------------------------------------------------------
# compute-dtype weight cache
# ----------------------------------------------------------------------------------------------
_WT_DTYPE = _np.dtype([("src", "u8"), ("dst_c", "u8"), ("dst_t", "u8"), ("rows", "i4"), ("cols", "i4"), ("tile0", "i4"), ("tiles_n", "i4")])


class WeightCache:
    """(cast copy [out,in], transposed cast copy [in,out]) of nn.Linear weights in the compute dtype, rebuilt when the fp32
    master changes (tensor version counter, storage pointer or FusedAdam's epoch).  Every cache registers itself; the first
    stale lookup after an optimizer step refreshes ALL registered caches of that (device, dtype) in one launch
    (maed_weight_refresh) into persistent buffers -- 32 matrices per step at cfg3 used to be 32 launches."""

    _registry = weakref.WeakSet()

    def __init__(self):
        self._key = None        # state the images were built from
        self._val = None        # what get() returns: [(compute-dtype [out,in] image or the fp32 master itself, transposed image)]
        self._images = None     # persistent buffers [(cast copy or None in f32 mode, transposed copy)]
        self._weights = None
        self._dtype = None
        WeightCache._registry.add(self)

    @staticmethod
    def _make_key(weights, dtype):
        return (WEIGHT_EPOCH, dtype, tuple((w.data_ptr(), w._version, tuple(w.shape)) for w in weights))

    def get(self, weights, dtype):
        if self._make_key(weights, dtype) != self._key:
            self._weights, self._dtype = list(weights), dtype
            WeightCache._registry.add(self)     # (copy.deepcopy / unpickling create caches without running __init__)
            WeightCache._refresh_group(self)
        return self._val

    @classmethod
    def _refresh_group(cls, me):
        dev, dtype = me._weights[0].device, me._dtype
        group = [c for c in cls._registry if c._weights is not None and c._dtype == dtype and c._weights[0].device == dev]
        rows, keep = [], []
        tile0 = 0
        with torch.no_grad():
            for c in group:
                ok = c._images is not None and len(c._images) == len(c._weights) and all(
                    im[1].shape == (w.shape[1], w.shape[0]) and im[1].dtype == dtype and im[1].device == dev for im, w in zip(c._images, c._weights))
                if not ok:      # (re)allocate the persistent images
                    c._images = [(None if dtype == torch.float32 else torch.empty(w.shape, dtype=dtype, device=dev),
                                  torch.empty(w.shape[1], w.shape[0], dtype=dtype, device=dev)) for w in c._weights]
                val = []
                for w, (wc, wt) in zip(c._weights, c._images):
                    src = _c(w.detach())
                    keep.append(src)
                    R, Cn = src.shape
                    tn = (Cn + 63) // 64
                    rows.append((_p(src), 0 if wc is None else _p(wc), _p(wt), R, Cn, tile0, tn))
                    tile0 += ((R + 63) // 64) * tn
                    val.append((src if dtype == torch.float32 else wc, wt))
                c._val = val
        for lo in range(0, len(rows), 256):     # WT_MAX_ENTRIES per launch
            chunk = rows[lo:lo + 256]
            base = chunk[0][5]
            tab = _np.array([r[:5] + (r[5] - base, r[6]) for r in chunk], dtype=_WT_DTYPE)
            ntiles = (chunk[-1][5] - base) + ((chunk[-1][3] + 63) // 64) * chunk[-1][6]
            tab_dev = _upload_table(tab, dev)
            check(L.lib().maed_weight_refresh(_p(tab_dev), len(chunk), ntiles, dt_code(dtype), _stream()), "weight_refresh")
        for c in group:
            c._key = cls._make_key(c._weights, dtype)


_SCRATCH = {}


def _aligned_bytes(nbytes, device, align=256):
    """uint8 buffer whose data pointer is `align`-byte aligned (the block kernels carve 256-B aligned sub-buffers out of it;
    the HIP caching allocator already aligns to 512 B, a host allocation -- tests/hostsim -- does not)"""
    buf = torch.empty(nbytes + align, dtype=torch.uint8, device=device)
    off = (-buf.data_ptr()) % align
    return buf[off:off + nbytes]


def twin_join():
    """the current stream waits for the cast passes twin forwards left on the library's side stream (maed_ste_block_twin_join): called where no backward is
    guaranteed to do it -- at the end of a chain of blocks, before a work buffer is replaced"""
    check(L.lib().maed_ste_block_twin_join(_stream()), "ste_block_twin_join")


def _scratch(nbytes, device, tag=None):
    key = (device.index, _stream(), tag)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None and isinstance(tag, tuple) and tag and tag[0] == "twin":
            twin_join()         # a pending cast pass may still read the buffer this one replaces
        buf = _aligned_bytes(nbytes, device)
        _SCRATCH[key] = buf
    return buf


# ----------------------------------------------------------------------------------------------
# autograd Functions
# ----------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = x W^T + b through maed_gemm_nt (nn.Linear / 1x1 conv semantics); x (M,K) in the compute dtype,
    W, b fp32 masters.  Returns y in the compute dtype."""

    @staticmethod
    def forward(ctx, x, weight, bias, cache, direct_w=None, direct_b=None):
        """direct_w / direct_b: the Parameters behind weight / bias (weight may be a view of direct_w) when this is their only use per forward: see direct_grad_slot"""
        (wc, wt), = cache.get([weight], x.dtype)
        ctx.save_for_backward(x, weight, bias)
        ctx.wt = wt
        ctx.direct = None
        if direct_w is not None and LinearFn._grad_at_apply and any(ctx.needs_input_grad[1:3]):
            ctx.direct = (direct_w, direct_b)
            direct_grad_begin(direct_w, direct_b)
        x32 = shadow_of(x)
        if x32 is not None:     # bf16 graph over fp32 shadows: the product on the fp32 operands (fp32 master weight, split-bf16 engine), an fp32 result; the backward stays bf16
            return gemm_nt(_c(x32), _c(weight.detach()), L.EPI_STORE, bias=bias)
        return gemm_nt(_c(x), wc, L.EPI_STORE, bias=bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        dy = _c(dy if dy.dtype == x.dtype else dy.to(x.dtype))
        if lib_matmul_dtype(dy.dtype) and weight.shape[0] % 8 == 0 and weight.shape[1] % 8 == 0:
            if ctx.direct is not None:
                gw, gb = direct_grad_slot(ctx.direct[0]), direct_grad_slot(ctx.direct[1])
                if gw is not None and (bias is None or gb is not None):
                    gemm_tn_wgrad(dy, x, dW=gw.view(weight.shape), dbias=gb)
                    dx = gemm_nt(dy, ctx.wt, L.EPI_STORE) if ctx.needs_input_grad[0] else None
                    direct_grad_done(*ctx.direct)
                    return dx, None, None, None, None, None
                direct_grad_cancel(*ctx.direct)     # (counted in, taken by autograd after all: AccumulateGrad's own hook reports it)
            db = torch.zeros_like(bias) if bias is not None else None
            dW = gemm_tn_wgrad(dy, x, dbias=db)
            dx = gemm_nt(dy, ctx.wt, L.EPI_STORE) if ctx.needs_input_grad[0] else None
            return dx, dW, db, None, None, None
        if ctx.direct is not None:
            direct_grad_cancel(*ctx.direct)
        db = torch.zeros_like(bias) if bias is not None else None
        dyt, _ = transpose_cast(dy, dy.dtype, colsum=db)
        xt, _ = transpose_cast(x, x.dtype)
        tiles = max(1, (weight.shape[0] // 128) * (weight.shape[1] // 128))
        dW = gemm_nt(dyt, xt, L.EPI_ATOMIC_F32, splitk=max(1, min(dyt.shape[1] // 256, 1024 // tiles)))
        dx = gemm_nt(dy, ctx.wt, L.EPI_STORE) if ctx.needs_input_grad[0] else None
        return dx, dW, db, None, None, None

    _grad_at_apply = True

    @classmethod
    def apply(cls, *args):
        LinearFn._grad_at_apply = torch.is_grad_enabled()
        return super().apply(*args)


class EmbedAddFn(torch.autograd.Function):
    """tokens = cat(cls, patch) + pos_embed + temp_embed[:, :T]  (vision_transformer.py:392-399)"""

    @staticmethod
    def forward(ctx, patch, cls, pos, temp, T):
        F_, Pm1, C_ = patch.shape
        tok = torch.empty(F_, Pm1 + 1, C_, dtype=torch.float32, device=patch.device)
        check(L.lib().maed_embed_add_fwd(_p(_c(patch)), dt_code(patch.dtype), _p(_c(cls)), _p(_c(pos)), _p(_c(temp)), _p(tok),
                                         F_, Pm1 + 1, C_, T, _stream()), "embed_add_fwd")
        ctx.T, ctx.pdtype = T, patch.dtype
        ctx.temp_shape = temp.shape
        return tok

    @staticmethod
    def backward(ctx, dtok):
        dtok = _c(dtok)
        F_, P, C_ = dtok.shape
        dpatch = torch.empty(F_, P - 1, C_, dtype=ctx.pdtype, device=dtok.device)
        dpos = torch.zeros(1, P, C_, dtype=torch.float32, device=dtok.device)
        dtemp = torch.zeros(ctx.temp_shape, dtype=torch.float32, device=dtok.device)       # (1, max_seqlen, 1, C): rows t < T are the first T * C floats
        check(L.lib().maed_embed_add_bwd(_p(dtok), _p(dpatch), dt_code(ctx.pdtype), _p(dpos), _p(dtemp), F_, P, C_, ctx.T, _stream()), "embed_add_bwd")
        dcls = dpos[:, :1].clone()                      # sum_f dtok[f][0]: the token-0 row of the positional sum (no second reduction over dtok)
        return dpatch, dcls, dpos, dtemp, None



# (weakref to the residual-gradient tensor a Block backward returned, its compute-dtype copy).  The next Block backward
# uses the copy only if it receives THAT VERY tensor object as grad_output (identity, not data_ptr: allocator reuse).
_TWIN = [None, None, None]      # [weakref to dx, its compute-dtype copy, chain index of the block that left it (Block._chain_index, None if unknown)]
TWIN_HITS = [0, 0]   # [hits, misses] -- diagnostics
_TWIN_WARNED = [False]


# ---- parameter gradients written straight into .grad by a backward kernel, without autograd's AccumulateGrad ------------------------------------------------
# An autograd Function that returns dW costs, per parameter and step, a zero-fill (the kernels accumulate with fp32 atomics), the kernel, and AccumulateGrad's add
# into .grad: 4 tiny framework launches per nn.Linear (54 framework launches per step).  At call sites where the parameter has ONE use per forward
# (the decoder head's fc1 / fc2, pre_logits, the encoder's final LayerNorm, the patch projection) the backward accumulates into .grad itself -- when there is an fp32
# .grad to accumulate into (ParamArena's views, or any optimizer that keeps gradients allocated) -- and tells the data-parallel bucketer through the hook the bucketer
# left on the parameter.  A forward counts itself in (two forwards before one backward: trainer.py:253-262), the last backward reports.
DIRECT_GRADS = os.environ.get("MAED_DIRECT_GRADS", "1") == "1"


def direct_grad_slot(p):
    """the fp32 tensor a backward kernel may ACCUMULATE p's gradient into, or None (autograd's way)"""
    if not DIRECT_GRADS or p is None or not p.is_leaf:
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or g.shape != p.shape or g.device != p.device or not g.is_contiguous():
        return None
    return g


def direct_grad_begin(*params):
    for p in params:
        if p is not None:
            p._maed_direct = getattr(p, "_maed_direct", 0) + 1


def direct_grad_cancel(*params):
    """counted in, but the gradient goes through autograd after all (no .grad to accumulate into): AccumulateGrad's own hook reports it"""
    for p in params:
        if p is not None:
            p._maed_direct = max(getattr(p, "_maed_direct", 1) - 1, 0)


def direct_grad_done(*params):
    """the backward that was counted in by direct_grad_begin has written its share: the parameter's last one reports it final"""
    for p in params:
        if p is None:
            continue
        n = getattr(p, "_maed_direct", 1) - 1
        p._maed_direct = max(n, 0)
        hook = getattr(p, "_maed_ready", None)
        if n <= 0 and hook is not None:
            hook(p)


class ReportingFn(torch.autograd.Function):
    """Base of the Functions whose backward writes parameter gradients itself and reports through owner.grads_ready: the owner's
    _pending_backwards counts forwards whose backward WILL run.  Inside Function.forward grad mode is always off and
    ctx.needs_input_grad ignores torch.no_grad(), so the mode is sampled at apply() time."""
    _grad_mode_at_apply = True

    @classmethod
    def apply(cls, *args):
        ReportingFn._grad_mode_at_apply = torch.is_grad_enabled()
        return super().apply(*args)

    @staticmethod
    def will_run_backward(ctx):
        return ReportingFn._grad_mode_at_apply and any(ctx.needs_input_grad)


class STEBlockFn(ReportingFn):
    """One STE Block through maed_ste_block_{fwd,bwd}.  Parameter gradients are accumulated by the
    kernels directly into p.grad (fp32, allocated on demand); autograd sees None for them, and the
    module's `grads_ready` hook tells the data-parallel bucketer when they are final."""

    @staticmethod
    def forward(ctx, x, block, dims, H, T, *params):
        lib = L.lib()
        x = _c(x)
        d = L.BlockDims(*dims)
        y = torch.empty_like(x)
        # fp32 forward, bf16 twins for the backward (set_float32_backward_precision("bf16")): the arena has the bf16 block's layout, the backward its dims
        twin = block.compute_dtype == torch.float32 and bwd_twin() and ReportingFn.will_run_backward(ctx) and dims[7] != L.IMPL_VALU and dims[2] % 8 == 0 and dims[5] % 8 == 0
        if twin:
            dims = dims[:6] + (BF16,) + dims[7:]
            d16 = L.BlockDims(*dims)
            pr = block._c_params_masters()
            saved = _aligned_bytes(lib.maed_ste_block_saved_bytes(C.byref(d16)), x.device)
            # two work buffers, alternating: the cast pass of block i (side stream) reads buffer i % 2 while block i + 1 writes the other
            slot = getattr(block, "_chain_index", 0) & 1
            work = _scratch(lib.maed_ste_block_twin_work_bytes(C.byref(d)), x.device, tag=("twin", slot))
            check(lib.maed_ste_block_fwd_twin(C.byref(d), C.byref(pr), _p(x), _p(y), _p(saved), _p(work), slot, _stream()), "ste_block_fwd_twin")
            TWIN_FORWARDS[0] += 1
        else:
            pr = block._c_params(block.compute_dtype)
            saved = _aligned_bytes(lib.maed_ste_block_saved_bytes(C.byref(d)), x.device)
            # no backward will follow (torch.no_grad / nothing requires grad): the inference entry point leaves out what only the backward reads
            entry = lib.maed_ste_block_fwd if ReportingFn.will_run_backward(ctx) else lib.maed_ste_block_infer
            check(entry(C.byref(d), C.byref(pr), _p(x), _p(y), _p(saved), _stream()), "ste_block_fwd")
        ctx.block, ctx.dims, ctx.bdt = block, dims, (torch.bfloat16 if twin else block.compute_dtype)
        ctx.save_for_backward(x, saved)
        # a hand-off left by the LAST block of an earlier backward pass (nobody consumes block 0's) is stale once a new forward runs: dropping it here frees its tensor
        _TWIN[0], _TWIN[1], _TWIN[2] = None, None, None
        if ReportingFn.will_run_backward(ctx):        # a forward under no_grad has no backward to pair with
            block._pending_backwards += 1
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.lib()
        x, saved = ctx.saved_tensors
        block = ctx.block
        d = L.BlockDims(*ctx.dims)
        cdt = ctx.bdt           # dtype of the backward's operands: the block's compute dtype, or bf16 behind a twin forward
        pr = block._c_params(cdt, backward=True)
        gr = block._c_grads()
        dy = _c(dy)
        dx = torch.empty_like(dy)
        scratch = _scratch(lib.maed_ste_block_scratch_bytes(C.byref(d)), dy.device)
        # consecutive blocks hand the compute-dtype copy of the residual gradient along (no cast pass in between)
        tw_in = None
        if cdt != torch.float32:
            ref, cand, producer = _TWIN
            mine = getattr(block, "_chain_index", None)
            if ref is not None and ref() is dy and cand.shape == dy.shape and cand.dtype == cdt:
                tw_in = cand
            elif ref is not None and producer is not None and mine is not None and producer == mine + 1 and not _TWIN_WARNED[0]:
                # (only when the hand-off was meant for THIS block: the first block of a backward chain legitimately finds the previous chain's last one)
                # a block handed its compute-dtype gradient copy on, but the tensor that arrives is not the one it returned (a hook that clones or rescales
                # gradients, retain_graph replays, activation checkpointing ...): correct -- the copy is re-made from dy -- but a cast pass per block
                _TWIN_WARNED[0] = True
                import warnings
                warnings.warn("maed_amd: the residual-gradient hand-off between consecutive STE blocks was bypassed (the gradient tensor arriving at a block is not "
                              "the one the next block returned); falling back to one extra cast pass per block", RuntimeWarning, stacklevel=2)
            TWIN_HITS[0 if tw_in is not None else 1] += 1
        _TWIN[0], _TWIN[1], _TWIN[2] = None, None, None
        tw_out = torch.empty(dy.shape, dtype=cdt, device=dy.device) if cdt != torch.float32 else None
        check(lib.maed_ste_block_bwd(C.byref(d), C.byref(pr), C.byref(gr), _p(x), _p(dy), _p(dx), _p(saved), _p(scratch),
                                     _p(tw_in), _p(tw_out), _stream()), "ste_block_bwd")
        if tw_out is not None:
            _TWIN[0], _TWIN[1], _TWIN[2] = weakref.ref(dx), tw_out, getattr(block, "_chain_index", None)
        block._pending_backwards -= 1
        if block._pending_backwards == 0 and block.grads_ready is not None:
            block.grads_ready(block)
        return (dx, None, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 5)


# ----------------------------------------------------------------------------------------------
# backbone helpers: batched weight standardisation, fused GroupNorm(+residual)(+ReLU)
# ----------------------------------------------------------------------------------------------

_WS_DTYPE = _np.dtype([("w", "u8"), ("gw", "u8"), ("gout", "u8"), ("dst_off", "i8"), ("dst_t_off", "i8"), ("O", "i4"), ("I", "i4"), ("KHW", "i4"),
                       ("fstart", "i4"), ("gout_f32", "i4"), ("pad_", "i4")])


def _ws_table(weights, grads=None, gouts=None, transposed=None, gout_f32=None):
    """transposed: set of conv indices that also get a transposed image (offsets appended after the regular arena)"""
    tab = _np.zeros(len(weights), dtype=_WS_DTYPE)
    off = fstart = 0
    for i, w in enumerate(weights):
        O, I, kh, kw = w.shape
        tab[i] = (w.data_ptr(), 0 if grads is None else grads[i], 0 if gouts is None else gouts[i], off, -1, O, I, kh * kw, fstart,
                  0 if gout_f32 is None else int(gout_f32[i]), 0)
        off += w.numel()
        fstart += O
    total = off
    t_offs = {}
    for i in sorted(transposed or ()):
        tab[i]["dst_t_off"] = total
        t_offs[i] = total
        total += weights[i].numel()
    return tab, off, fstart, total, t_offs


class WeightStdFn(ReportingFn):
    """All StdConv2dSame weights of a backbone standardised in ONE launch (forward) / ONE launch (backward).
    Outputs are channels_last-strided (O,I,kh,kw) views of one arena in the compute dtype.  Like STEBlockFn
    the backward accumulates straight into p.grad and reports through owner.grads_ready.

    Convolutions listed in owner._gemm_convs (1x1, stride 1: they run on maed_gemm_nt / maed_gemm_tn_wgrad through
    Conv1x1Fn) additionally get the transposed (I, O) image (owner._w_std_t[i]) and hand their weight gradient over as
    an fp32 slice of owner._dw_arena instead of through autograd."""

    @staticmethod
    def forward(ctx, owner, dtype, eps, *weights):
        weights = [_c(w) for w in weights]
        # convolutions that run on the library's own kernels: they also get the transposed image and an fp32 dW slice
        direct = getattr(owner, "_direct_convs", None)
        gemm = set((direct if direct is not None else getattr(owner, "_gemm_convs", ())) or ()) if lib_matmul_dtype(dtype, getattr(owner, "f32_matmul", None)) else set()
        tab, _, nf, total, t_offs = _ws_table(weights, transposed=gemm)
        dev = weights[0].device
        out = torch.empty(total, dtype=dtype, device=dev)
        stats = torch.empty(nf * 2, dtype=torch.float32, device=dev)
        tab_dev = _upload_table(tab, dev)
        # transposed images by a tile transpose of the forward image (MAED_WS_TILED=0: element-wise strided stores from the statistics kernel, A/B knob)
        t_tiles = sum(-(-weights[i].shape[0] // 64) * -(-(weights[i].numel() // weights[i].shape[0]) // 64) for i in t_offs) if os.environ.get("MAED_WS_TILED", "1") == "1" else 0
        check(L.lib().maed_weight_std_fwd(_p(tab_dev), len(weights), nf, _p(out), dt_code(dtype), _p(stats), eps, t_tiles, _stream()), "weight_std_fwd")
        ctx.owner, ctx.dtype, ctx.eps, ctx.stats, ctx.nf = owner, dtype, eps, stats, nf
        ctx.weights = weights
        ctx.set_materialize_grads(False)     # the library convolutions hand their dW over in fp32 slices: no zero tensors in their place
        if ReportingFn.will_run_backward(ctx):
            owner._pending_backwards += 1
        views, off = [], 0
        for w in weights:
            O, I, kh, kw = w.shape
            views.append(out[off:off + w.numel()].view(O, kh, kw, I).permute(0, 3, 1, 2))
            off += w.numel()
        # (I, O) for the 1x1 convolutions, (9*I, O) = (3,3,I,O) for the 3x3 ones
        owner._w_std_t = {i: out[o:o + weights[i].numel()].view(-1, weights[i].shape[0]) for i, o in t_offs.items()}
        # fp32 weight-gradient arena of the GEMM convolutions (maed_gemm_tn_wgrad accumulates with atomics: zero it once per step)
        owner._dw_arena, owner._dw_slices = None, {}
        if gemm and ReportingFn.will_run_backward(ctx):
            n = sum(weights[i].numel() for i in gemm)
            # ONE arena per owner, re-used from step to step (zero-filled here, behind the previous step's weight_std_bwd in stream order): the table of
            # maed_weight_std_bwd then holds the same pointers every step (no upload, ops._upload_table; a captured step replays it).  A second forward before the
            # first one's backward (trainer.py's two-forward step) gets an arena of its own, as before.
            keep = getattr(owner, "_dw_arena_persist", None)
            if owner._pending_backwards == 1 and keep is not None and keep.numel() == n and keep.device == dev:
                owner._dw_arena = keep.zero_()
            else:
                owner._dw_arena = torch.zeros(n, dtype=torch.float32, device=dev)
                if owner._pending_backwards == 1:
                    owner._dw_arena_persist = owner._dw_arena
            o = 0
            for i in sorted(gemm):
                owner._dw_slices[i] = owner._dw_arena[o:o + weights[i].numel()].view(weights[i].shape[0], -1)     # (O, I) / (O, 9*I)
                o += weights[i].numel()
        ctx.dw_slices = owner._dw_slices
        return tuple(views)

    @staticmethod
    def backward(ctx, *gouts):
        """Runs on the caller's stream after joining the side stream (the fp32 dW slices and the GroupNorm dgamma/dbeta sums are written there) --
        except for a group the owner marked `_ws_on_side` (per-stage mode: every stage but the one whose backward runs last): those enqueue
        their kernel AND their readiness report on the side stream itself, behind that stage's weight gradients, so the caller's stream -- the
        dy -> dx chain of the next stage -- never waits for them; the last group's join covers them."""
        dev = ctx.weights[0].device
        gn_affine_flush(dev)        # the GroupNorm layers' deferred dgamma / dbeta sums: on the caller's stream, before anything below hands the gradients on
        st = _SIDE.get(dev) if (_side_on() and getattr(ctx.owner, "_ws_on_side", False)) else None
        if st is None or not st[1]:
            side_stream_join(dev)
            return WeightStdFn._backward_body(ctx, gouts)
        side = st[0]
        side.wait_stream(torch.cuda.current_stream(dev))     # gradients autograd carried here (MIOpen convolutions), p.grad zero-fills
        _keep_until_join(dev, st, [g for g in gouts if g is not None])
        with torch.cuda.stream(side):
            return WeightStdFn._backward_body(ctx, gouts)

    @staticmethod
    def _backward_body(ctx, gouts):
        weights, owner = ctx.weights, ctx.owner
        params = owner.conv_weights()
        keep, gptr, optr, f32 = [], [], [], []
        for i, (w, p, g) in enumerate(zip(weights, params, gouts)):
            direct = ctx.dw_slices.get(i)
            if g is None and direct is None:
                gptr.append(0); optr.append(0); f32.append(0)
                continue
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            if direct is not None:      # fp32 dW written by Conv1x1Fn.backward (autograd carried nothing for this conv)
                gptr.append(p.grad.data_ptr()); optr.append(direct.data_ptr()); f32.append(1)
                continue
            g = g.to(ctx.dtype)
            g = g.contiguous(memory_format=torch.channels_last) if w.shape[2] * w.shape[3] > 1 else g.contiguous()
            keep.append(g)
            gptr.append(p.grad.data_ptr()); optr.append(g.data_ptr()); f32.append(0)
        tab, _, nf, _, _ = _ws_table(weights, gptr, optr, gout_f32=f32)
        tab_dev = _upload_table(tab, weights[0].device)
        check(L.lib().maed_weight_std_bwd(_p(tab_dev), len(weights), nf, dt_code(ctx.dtype), _p(ctx.stats), ctx.eps, _stream()), "weight_std_bwd")
        owner._pending_backwards -= 1
        if owner._pending_backwards == 0 and owner.grads_ready is not None:
            owner.grads_ready(owner)
        return (None, None, None) + (None,) * len(weights)


# residual gradients handed on unmasked: data_ptr of the tensor -> (weakref to it, the ReLU bit mask that still has to be applied); consumed (popped) by
# Conv1x1Fn.backward.  Keyed by address AND identity (the weakref must still point at the tensor that arrives).
LAZY_RES = {}


GN_SYNC_WORDS = 80      # MAED_GN_SYNC_WORDS (include/maed_hip.h): 4-byte words per frame of maed_groupnorm_bwd's frame_sync scratch
GN_DEFER_AFFINE = os.environ.get("MAED_GN_DEFER_AFFINE", "1") == "1"      # A/B knob: 0 = one closing kernel per GroupNorm layer (rounds 2-5)
_GN_DEFERRED = {}       # device -> [(ab, dgamma, dbeta, N, C)]: GroupNorm backward partials whose column sums are still to be folded into the gradients


def gn_affine_defer(device, ab, dgamma, dbeta, N, C_):
    pend = _GN_DEFERRED.setdefault(device, [])
    if not pend:
        # safety net: a pass whose weight standardisation has no backward (frozen convolutions) still folds its partials when the engine finishes
        try:
            torch.autograd.Variable._execution_engine.queue_callback(lambda: gn_affine_flush(device))
        except RuntimeError:        # not inside a backward pass (a Function called by hand): the caller flushes
            pass
    pend.append((ab, dgamma, dbeta, N, C_))


def gn_affine_flush(device):
    """fold the deferred GroupNorm partials of `device` into dgamma / dbeta: one launch on the CURRENT stream (which is the stream the GroupNorm backward kernels ran
    on, or one ordered behind it)"""
    pend = _GN_DEFERRED.get(device)
    if not pend:
        return
    items = (L.GnAffineItem * len(pend))()
    for i, (ab, dg, db, N, C_) in enumerate(pend):
        items[i].ab, items[i].dgamma, items[i].dbeta, items[i].N, items[i].C = _p(ab), _p(dg), _p(db), N, C_
    check(L.lib().maed_gn_affine_grad_batch(C.cast(items, C.c_void_p), len(pend), _stream()), "gn_affine_grad_batch")
    pend.clear()


class GroupNormFn(torch.autograd.Function):
    """y = act(GroupNorm32(x) * gamma + beta [+ residual]) on channels_last tensors (maed_groupnorm_fwd/bwd).
    direct=True: gamma/beta gradients are accumulated by the kernel straight into gamma.grad / beta.grad (the
    owner module reports them through its grads_ready callback) instead of travelling through autograd."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, eps, relu, direct, sums=None, ab=None, stats_ready=False, lazy_res=False, sync=None):
        """sums (N,32,2) f64 / ab (N,C,2) f32: optional PRE-ZEROED scratch slices (ResNetV2 zeroes one arena per pass for all
        its 52 layers instead of one memset per layer and direction).  stats_ready: `sums` already holds the statistics of x (the
        producing convolution's epilogue accumulated them: Conv1x1Fn / Conv3x3Fn gn_sums) -- no statistics pass."""
        N, C_, H, W = x.shape
        x32 = shadow_of(x)                       # bf16 graph over fp32 shadows (shadow_put): the forward runs on the shadows, the twins are what is saved
        r32 = shadow_of(residual) if x32 is not None else None
        if x32 is not None and not x.is_contiguous(memory_format=torch.channels_last):
            twin_fill(x)                         # an unfilled twin in another layout would be COPIED below, unfilled: fill it first (the kernel then need not)
        assert x32 is None or residual is None or r32 is not None, "GroupNormFn: the residual of a shadowed input has no shadow"
        x = x.contiguous(memory_format=torch.channels_last)
        if residual is not None:
            residual = residual.contiguous(memory_format=torch.channels_last).to(x.dtype)
        xin, rin = (x, residual) if x32 is None else (x32.contiguous(memory_format=torch.channels_last), r32)
        y = torch.empty_like(xin, memory_format=torch.channels_last)
        zeroed = sums is not None
        if sums is None:
            sums = torch.empty(N, 32, 2, dtype=torch.float64, device=x.device)
        ctx.has_res = residual is not None
        # ReLU after a residual add: the backward cannot recompute the mask from x -> 1 bit per element instead of re-reading y
        need_mask = relu and ctx.has_res and (x.requires_grad or residual.requires_grad or gamma.requires_grad)
        mask = torch.empty(N * H * W * (C_ // 8), dtype=torch.uint8, device=x.device) if need_mask else None
        if x32 is not None:
            # the pass that reads the fp32 convolution output and writes the fp32 result also writes both bf16 twins: x's (handed out unfilled by the convolution:
            # twin_later) and its own result's
            fill_x = x.data_ptr() in _UNFILLED
            y16 = torch.empty_like(y, dtype=torch.bfloat16)
            twin_fill(residual)
            check(L.lib().maed_groupnorm_fwd_twin(_p(xin), _p(rin), _p(gamma), _p(beta), _p(y), _p(sums), _p(mask), N, H * W, C_, eps, int(relu),
                                                  2 if (stats_ready and zeroed) else int(zeroed), _p(x) if fill_x else None, _p(y16), _stream()), "groupnorm_fwd_twin")
            _UNFILLED.discard(x.data_ptr())
            y = shadow_put(y16, y)
        else:
            check(L.lib().maed_groupnorm_fwd(_p(xin), _p(rin), _p(gamma), _p(beta), _p(y), _p(sums), _p(mask), N, H * W, C_, eps, int(relu),
                                             dt_code(xin.dtype), 2 if (stats_ready and zeroed) else int(zeroed), _stream()), "groupnorm_fwd")
        ctx.ab = ab
        ctx.sync = sync if ab is not None else None      # N * GN_SYNC_WORDS zero words (any 4-byte dtype): frame_sync of the one-pass backward, single use like ab
        ctx.save_for_backward(x, mask, sums)
        ctx.eps, ctx.relu, ctx.direct = eps, relu, direct
        # lazy_res: the residual's gradient (dy masked by the ReLU bits) is not materialised -- backward hands dy itself on and registers the bit mask
        # for it; the consumer (Conv1x1Fn.backward of the block's conv1, which adds the shortcut gradient inside its input-gradient GEMM) applies it
        ctx.lazy_res = bool(lazy_res) and need_mask
        ctx.gamma, ctx.beta = gamma, beta   # parameters (leaf tensors): kept by reference for .grad access
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mask, sums = ctx.saved_tensors
        gamma, beta = ctx.gamma, ctx.beta
        N, C_, H, W = x.shape
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        dres = torch.empty_like(x, memory_format=torch.channels_last) if (ctx.has_res and not ctx.lazy_res) else None
        if ctx.direct:
            if gamma.grad is None:
                gamma.grad = torch.zeros_like(gamma)
            if beta.grad is None:
                beta.grad = torch.zeros_like(beta)
            dgamma, dbeta = gamma.grad, beta.grad
        else:
            dgamma = torch.zeros(C_, dtype=torch.float32, device=x.device)
            dbeta = torch.zeros(C_, dtype=torch.float32, device=x.device)
        ab, ab_zeroed, sync = ctx.ab, ctx.ab is not None, ctx.sync
        ctx.ab = ctx.sync = None                        # single use: a second backward through this node gets fresh scratch
        if ab is None:
            ab = torch.empty(N, C_, 2, dtype=torch.float32, device=x.device)
        if sync is None:
            sync = torch.zeros(N * GN_SYNC_WORDS, dtype=torch.int32, device=x.device)
        # kernel-written dgamma / dbeta are read only after WeightStdFn.backward: with the pass's scratch arena (ab_zeroed) the closing column sum over the frames is
        # DEFERRED -- the layer's partials stay in `ab` and ONE maed_gn_affine_grad_batch launch folds every layer of the pass (gn_affine_flush, called by
        # WeightStdFn.backward and by the engine's end-of-pass callback): 52 launches of 6 us and 52 stream fences per step less (round 6).  Without the arena:
        # the per-layer closing kernel on the side stream, as before.
        defer = ctx.direct and ab_zeroed and GN_DEFER_AFFINE
        aux = side_stream_handle(x.device, ab) if (ctx.direct and not defer) else None
        check(L.lib().maed_groupnorm_bwd(_p(x), _p(mask), _p(dy), _p(sums), _p(gamma), _p(beta), _p(dx), _p(dres), None if defer else _p(dgamma),
                                         None if defer else _p(dbeta), _p(ab), N, H * W, C_, ctx.eps, int(ctx.relu), dt_code(x.dtype), int(ab_zeroed), _p(sync), aux,
                                         _stream()), "groupnorm_bwd")
        if defer:
            gn_affine_defer(x.device, ab, dgamma, dbeta, N, C_)
        if ctx.lazy_res:
            for k in [k for k, (r, _) in LAZY_RES.items() if r() is None]:      # announcements whose consumer never ran (an interrupted backward)
                del LAZY_RES[k]
            LAZY_RES[dy.data_ptr()] = (weakref.ref(dy), mask)
            dres = dy
        if ctx.direct:
            return dx, dres, None, None, None, None, None, None, None, None, None, None
        return dx, dres, dgamma, dbeta, None, None, None, None, None, None, None, None


def stem_input(x, dtype, k, s, own=False):
    """fp32 NCHW clip frames -> compute dtype, channels_last, TF-SAME padded for a kernel-k / stride-s convolution, in ONE pass (maed_stem_input); returns the
    padded tensor as an (N, C, Hp, Wp) channels_last view: the convolution then runs with padding 0.
    own=True: the layout of maed_stem7x7s2_* instead -- 4 channel slots per pixel (the 4th zero) and one more zero column on the right (even width)"""
    N, C_, H, W = x.shape
    ph = max((math.ceil(H / s) - 1) * s + k - H, 0)
    pw = max((math.ceil(W / s) - 1) * s + k - W, 0)
    cs, extra = (4, 1) if own else (C_, 0)
    y = torch.empty(N, H + ph, W + pw + extra, cs, dtype=dtype, device=x.device)
    check(L.lib().maed_stem_input(_p(x), _p(y), N, C_, H, W, ph // 2, ph - ph // 2, pw // 2, pw - pw // 2 + extra, cs, dt_code(dtype), _stream()), "stem_input")
    return y.permute(0, 3, 1, 2)


def stem7x7s2_supported(H, W, F_=1):
    """geometry AND clip size: the stem kernels address with 32-bit byte offsets (csrc/stem.hip STEM_CHECK_GEOM) -- a clip past those limits (about 2 675 frames of
    224 x 224) takes the vendor convolution instead of failing with MAED_ERR_SHAPE"""
    H, W, F_ = int(H), int(W), int(F_)
    if F_ < 1 or F_ * (H + 5) * (W + 6) * 8 >= 1 << 31 or F_ * (H // 2) * (W // 2) * 128 >= 1 << 32:
        return False
    return bool(L.lib().maed_stem7x7s2_supported(H, W))


class StemConvFn(torch.autograd.Function):
    """The stem convolution StdConv2dSame(3 -> 64, 7, stride 2) (resnetv2.py:74-93, :330-333) on maed_stem7x7s2_fwd / _wgrad.
    xp: the padded 4-slot image of stem_input(own=True) as an (F, 4, H+5, W+6) channels_last view; w: the standardised weight (64, I, 7, 7) channels_last view of
    WeightStdFn's arena; dw: its fp32 gradient slice (64, 147) (accumulated with atomics: WeightStdFn zeroes the arena once per step); sums: the (F, 32, 2) fp64
    statistics slice of the GroupNorm behind, or None.  The frames get no gradient."""

    @staticmethod
    def forward(ctx, xp, w, dw, sums, hw):
        H, W = hw
        F_ = xp.shape[0]
        assert xp.shape[1] == 4 and xp.shape[2] == H + 5 and xp.shape[3] == W + 6 and xp.dtype == torch.bfloat16 and w.shape[0] == 64 and w.shape[2:] == (7, 7), (xp.shape, w.shape)
        wc = w.permute(0, 2, 3, 1)
        assert wc.is_contiguous() and wc.shape[3] == 3, "stem: the standardised weight must be the channels_last (64, 7, 7, 3) image"
        ctx.save_for_backward(xp)
        ctx.dw, ctx.hw = dw, hw
        x32 = shadow_of(xp)
        if x32 is not None:     # bf16 graph over fp32 shadows: x32 is the TF-SAME padded fp32 image (stem_input own=False); the vendor's fp32 convolution as in the f32 modes
            assert sums is None, "stem: the GroupNorm behind a shadowed stem computes its own statistics"
            w32 = shadow_of(w)
            # (channels_last whatever the vendor solver returns: the GroupNorm behind writes this tensor's twin in place, by address)
            return twin_later(torch.nn.functional.conv2d(x32, w32, None, 2, 0).contiguous(memory_format=torch.channels_last))
        y = torch.empty((F_, 64, H // 2, W // 2), dtype=xp.dtype, device=xp.device, memory_format=torch.channels_last)
        wimg = torch.empty(64 * 224, dtype=xp.dtype, device=xp.device)
        check(L.lib().maed_stem7x7s2_fwd(_p(xp), _p(wc), _p(wimg), _p(y), _p(sums), F_, H, W, dt_code(xp.dtype), _stream()), "stem7x7s2_fwd")
        return y

    @staticmethod
    def backward(ctx, dy):
        (xp,) = ctx.saved_tensors
        H, W = ctx.hw
        dy = dy.contiguous(memory_format=torch.channels_last)
        assert ctx.dw is not None, "stem: no fp32 weight-gradient slice (WeightStdFn hands it out when the backward will run)"
        dw = ctx.dw
        scratch = torch.empty(L.lib().maed_stem7x7s2_wgrad_scratch_floats(xp.shape[0], H, W), dtype=torch.float32, device=dy.device)
        side_stream_run(lambda: check(L.lib().maed_stem7x7s2_wgrad(_p(dy), _p(xp), _p(dw), _p(scratch), xp.shape[0], H, W, dt_code(dy.dtype), _stream()), "stem7x7s2_wgrad"),
                        dy, xp, dw, scratch)
        return None, None, None, None, None


class MaxPool3s2SameFn(torch.autograd.Function):
    """MaxPool2dSame(3, 2) on a channels_last tensor (maed_maxpool3s2_same_fwd/bwd): no -inf padded copy, gather backward"""

    @staticmethod
    def forward(ctx, x):
        N, C_, H, W = x.shape
        x32 = shadow_of(x)
        x = (x if x32 is None else x32).contiguous(memory_format=torch.channels_last)
        Ho, Wo = (H + 1) // 2, (W + 1) // 2
        y = torch.empty((N, C_, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        idx = torch.empty(N * Ho * Wo * C_, dtype=torch.uint8, device=x.device)
        check(L.lib().maed_maxpool3s2_same_fwd(_p(x), _p(y), _p(idx), N, H, W, C_, dt_code(x.dtype), _stream()), "maxpool3s2_same_fwd")
        ctx.save_for_backward(idx)
        ctx.geom = (N, C_, H, W)
        return y if x32 is None else twin_of(y)

    @staticmethod
    def backward(ctx, dy):
        idx, = ctx.saved_tensors
        N, C_, H, W = ctx.geom
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty((N, C_, H, W), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
        check(L.lib().maed_maxpool3s2_same_bwd(_p(dy), _p(idx), _p(dx), N, H, W, C_, dt_code(dy.dtype), _stream()), "maxpool3s2_same_bwd")
        return dx


class Conv1x1Fn(torch.autograd.Function):
    """1x1 stride-1 convolution on a channels_last bf16 tensor as GEMMs on libmaed_hip (33 of the backbone's 53
    convolutions): y[(n,h,w), o] = sum_i x[(n,h,w), i] w[o, i].  Forward = maed_gemm_nt on the standardised weight (O, I),
    input gradient = maed_gemm_nt on its transposed image (I, O) (both written by the batched weight-standardisation
    kernel), weight gradient = maed_gemm_tn_wgrad accumulating in fp32 straight into the slice the weight-standardisation
    backward reads -- no output zero-fill, no fp32-workspace cast passes, no transposed activation copies.
    Measured against MIOpen's asm implicit-GEMM solvers at cfg3 (profiles/r01_conv1x1_micro.txt, r01_wgrad_micro.txt):
    forward 1.01 vs 2.33 ms, input gradient 1.02 vs 1.88 ms, weight gradient 1.47 vs 2.63 ms per step."""

    @staticmethod
    def forward(ctx, x, w, wt, dw, fork=False, gn_sums=None, stride=1, lazy_short=False, prec=None):
        """gn_sums (optional, pre-zeroed (N,32,2) f64): GroupNorm statistics of the output, accumulated by the GEMM's epilogue.
        x (N,I,H,W) channels_last; w (O,I,1,1) standardised weight (an output of WeightStdFn: the autograd edge orders
        its backward after ours); wt (I,O) transposed image; dw (O,I) fp32 accumulator (None when no gradient is wanted).
        fork=True additionally returns an alias of x for the block's identity shortcut: its gradient then arrives HERE and is
        added inside the input-gradient GEMM's epilogue (MAED_EPI_ADD) instead of by a separate autograd accumulation kernel.
        stride=2 (the downsample shortcuts of stages 2 and 3): the pixels the convolution reads are packed first (maed_subsample2_fwd),
        all three GEMMs then run on a quarter of the rows and the input gradient is spread back in one write pass."""
        N, I, H, W = x.shape
        x32, w32 = shadow_of(x), shadow_of(w)   # bf16 graph over fp32 shadows (shadow_put): the product runs on the fp32 operands, the bf16 operands are what is saved
        assert (x32 is None) == (w32 is None), "Conv1x1Fn: activation and weight must both (or neither) have fp32 shadows"
        x = x.contiguous(memory_format=torch.channels_last)
        O = w.shape[0]
        Ho, Wo = ((H + 1) // 2, (W + 1) // 2) if stride == 2 else (H, W)

        def rows_of(t):
            if stride == 2:
                assert not fork
                A_ = torch.empty(N * Ho * Wo, I, dtype=t.dtype, device=t.device)
                check(L.lib().maed_subsample2_fwd(_p(t), _p(A_), N, H, W, I, dt_code(t.dtype), _stream()), "subsample2_fwd")
                return A_
            assert stride == 1
            return t.permute(0, 2, 3, 1).reshape(N * H * W, I)

        def product(A_, w_):
            w2 = w_.reshape(O, I)
            w2 = w2 if w2.is_contiguous() else w2.contiguous()
            if gn_sums is not None:
                y_ = torch.empty(N * Ho * Wo, O, dtype=A_.dtype, device=A_.device)
                check(L.lib().maed_conv1x1_fwd(_p(A_), A_.stride(0), _p(w2), w2.stride(0), N * Ho * Wo, O, I, _p(y_), O, Ho * Wo, _p(gn_sums), mm_code(A_.dtype, prec),
                                               _stream()), "conv1x1_fwd")
                return y_
            return gemm_nt(A_, w2, L.EPI_STORE, prec=prec)

        A = rows_of(x)
        if x32 is not None:
            y32 = product(rows_of(x32.contiguous(memory_format=torch.channels_last)), w32).view(N, Ho, Wo, O).permute(0, 3, 1, 2)
            y = twin_later(y32)
        else:
            y = product(A, w).view(N, Ho, Wo, O).permute(0, 3, 1, 2)
        ctx.save_for_backward(A, wt)
        ctx.dw, ctx.geom, ctx.stride, ctx.prec = dw, (N, I, H, W, O, Ho, Wo), stride, prec
        ctx.lazy_short = bool(lazy_short) and fork      # the shortcut's gradient will arrive unmasked, its ReLU bits registered in LAZY_RES
        ctx.set_materialize_grads(False)
        if not fork:
            return y
        xa = x.view_as(x)
        if x32 is not None:
            shadow_put(xa, x32)     # (same address as x: the entry now holds the alias -- x itself stays alive as long as the alias does)
        return y, xa

    @staticmethod
    def backward(ctx, dy, g_short=None):
        A, wt = ctx.saved_tensors
        N, I, H, W, O, Ho, Wo = ctx.geom
        dx = None
        mask = None
        if ctx.lazy_short and g_short is not None:
            ent = LAZY_RES.pop(g_short.data_ptr(), None)
            if ent is None or ent[0]() is not g_short:
                raise RuntimeError("Conv1x1Fn: the shortcut gradient was announced as lazily masked (GroupNormFn lazy_res) but arrived as another tensor -- "
                                   "the residual must feed exactly one GroupNorm")
            mask = ent[1]
        if dy is None:              # only the shortcut carried a gradient
            if mask is not None:
                raise RuntimeError("Conv1x1Fn: lazily masked shortcut gradient without a gradient for the convolution itself")
            return g_short, None, None, None, None, None, None, None, None
        Y = dy.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(N * Ho * Wo, O)
        if ctx.needs_input_grad[0]:
            if g_short is not None:
                G = g_short.contiguous(memory_format=torch.channels_last).to(Y.dtype).permute(0, 2, 3, 1).reshape(N * H * W, I)
                dx = gemm_nt(Y, wt, L.EPI_ADD, aux=G, out2=mask, prec=bwd_prec(ctx.prec))
            else:
                dx = gemm_nt(Y, wt, L.EPI_STORE, prec=bwd_prec(ctx.prec))
            if ctx.stride == 2:
                g, dx = dx, torch.empty(N * H * W, I, dtype=dx.dtype, device=dx.device)
                check(L.lib().maed_subsample2_bwd(_p(g), _p(dx), N, H, W, I, dt_code(dx.dtype), _stream()), "subsample2_bwd")
            dx = dx.view(N, H, W, I).permute(0, 3, 1, 2)
        if ctx.dw is not None:
            dw, prec = ctx.dw, ctx.prec
            side_stream_run(lambda: gemm_tn_wgrad(Y, A, dW=dw, prec=bwd_prec(prec)), Y, A, dw)
        if mask is not None and dx is None:
            raise RuntimeError("Conv1x1Fn: lazily masked shortcut gradient but no input gradient requested")
        return dx, None, None, None, None, None, None, None, None




_ZERO_PAGE = {}


def _zero_page(device):
    """>= 128 bytes of zeros: the source of every out-of-image tap of the implicit-GEMM 3x3 convolution"""
    key = str(device)
    if key not in _ZERO_PAGE:
        _ZERO_PAGE[key] = _aligned_bytes(256, device).zero_()
    return _ZERO_PAGE[key]


def conv3x3(x, w_taps, stride=1, add=None, w_layout=0, gn_sums=None, prec=None):
    """y = conv3x3_SAME(x, w) on channels_last bf16 tensors through maed_conv3x3_fwd.  x (N,Cin,H,W) channels_last,
    w_taps: storage (Cout, 3, 3, Cin) contiguous (w_layout 0), or the transposed image (3, 3, Cout, Cin) of the FORWARD convolution whose
    input gradient this call computes (w_layout 1).  TF-SAME padding from the input size (resnetv2.py:51-59)."""
    N, I, H, W = x.shape
    x = x.contiguous(memory_format=torch.channels_last)
    O = w_taps.numel() // (9 * I)
    Ho, Wo = -(-H // stride), -(-W // stride)
    ph, pw = max((Ho - 1) * stride + 3 - H, 0), max((Wo - 1) * stride + 3 - W, 0)
    y = torch.empty(N, O, Ho, Wo, dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    if add is not None:
        add = add.contiguous(memory_format=torch.channels_last)
    check(L.lib().maed_conv3x3_fwd(_p(x), _p(w_taps), _p(_zero_page(x.device)), _p(y), N, H, W, I, O, stride, ph // 2, pw // 2, Ho, Wo, _p(add),
                                   w_layout, mm_code(x.dtype, prec), _p(gn_sums), _stream()), "conv3x3_fwd")
    return y


def conv3x3_s2_dgrad(dy, wt, H, W, pad_top, pad_left):
    """dX (N,I,H,W) of the stride-2 3x3 SAME convolution from channels_last bf16 dy (N,O,Ho,Wo) and the transposed image wt (3,3,I,O) of the standardised forward
    weight (maed_conv3x3_s2_dgrad: one launch per parity class of the input pixel)"""
    N, O, Ho, Wo = dy.shape
    I = wt.numel() // (9 * O)
    dy = dy.contiguous(memory_format=torch.channels_last)
    dx = torch.empty(N, I, H, W, dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
    check(L.lib().maed_conv3x3_s2_dgrad(_p(dy), _p(wt), _p(_zero_page(dy.device)), _p(dx), N, H, W, I, O, pad_top, pad_left, Ho, Wo, dt_code(dy.dtype), _stream()),
          "conv3x3_s2_dgrad")
    return dx


_TAPMASKS = {}


def _tapmask(N, H, W, device):
    """per-pixel 9-bit 'tap inside the image' mask of the 3x3 weight-gradient kernel; depends only on the feature-map size: cached"""
    key = (N, H, W, str(device))
    if key not in _TAPMASKS:
        n = (N * H * W + 63) // 64 * 64
        m = _aligned_bytes(2 * n, device)
        check(L.lib().maed_conv3x3_tapmask(_p(m), N, H, W, _stream()), "conv3x3_tapmask")
        _TAPMASKS[key] = m
    return _TAPMASKS[key]


def conv3x3_wgrad(dy, x, out=None, prec=None):
    """fp32 dW (O, 3, 3, I) of the stride-1 3x3 SAME convolution from channels_last bf16 dy (N,O,H,W) and x (N,I,H,W); accumulates
    into `out` (any (O, 9*I)-sized contiguous fp32 tensor) when given"""
    N, I, H, W = x.shape
    O = dy.shape[1]
    dW = torch.zeros(O, 3, 3, I, dtype=torch.float32, device=x.device) if out is None else out
    if x.dtype == torch.bfloat16:
        nb = L.lib().maed_conv3x3_wgrad_rows64_scratch_floats(N, H, W, I, O)
        if nb > 0:      # 64 -> 64 channels: one image row per work item, per-workgroup partial results in scratch (no atomics on a 147 KB hot spot)
            scratch = torch.empty(nb, dtype=torch.float32, device=x.device)
            check(L.lib().maed_conv3x3_wgrad_rows64(_p(dy), _p(x), _p(dW), _p(scratch), N, H, W, dt_code(x.dtype), _stream()), "conv3x3_wgrad_rows64")
            _keep_alive_on_stream(scratch)
            return dW
    check(L.lib().maed_conv3x3_wgrad(_p(dy), _p(x), _p(_tapmask(N, H, W, x.device)), _p(_zero_page(x.device)), _p(dW), N, H, W, I, O,
                                     mm_code(x.dtype, prec), _stream()), "conv3x3_wgrad")
    return dW


_S2_TABLES = {}


def conv3x3_s2_wgrad(dy, x, pad_top, pad_left, out=None):
    """fp32 dW (O, 3, 3, I) of the stride-2 3x3 SAME convolution from channels_last bf16 dy (N,O,Ho,Wo) and x (N,I,H,W); accumulates into `out` when given.
    The per-output-pixel gather tables depend only on the geometry: cached."""
    N, I, H, W = x.shape
    O, Ho, Wo = dy.shape[1:]
    key = (N, H, W, Ho, Wo, pad_top, pad_left, str(x.device))
    if key not in _S2_TABLES:
        n = (N * Ho * Wo + 63) // 64 * 64
        mask, rows = _aligned_bytes(2 * n, x.device), _aligned_bytes(4 * n, x.device)
        check(L.lib().maed_conv3x3_s2_tables(_p(mask), _p(rows), N, H, W, pad_top, pad_left, Ho, Wo, _stream()), "conv3x3_s2_tables")
        _S2_TABLES[key] = (mask, rows)
    mask, rows = _S2_TABLES[key]
    dW = torch.zeros(O, 3, 3, I, dtype=torch.float32, device=x.device) if out is None else out
    check(L.lib().maed_conv3x3_s2_wgrad(_p(dy), _p(x), _p(mask), _p(rows), _p(_zero_page(x.device)), _p(dW), N, H, W, I, O, Ho, Wo, dt_code(x.dtype), _stream()),
          "conv3x3_s2_wgrad")
    return dW


class Conv3x3Fn(torch.autograd.Function):
    """StdConv2dSame 3x3 (resnetv2.py:74-93) on the library's implicit-GEMM kernel: forward for any stride, input gradient for
    stride 1 (the same kernel on dY), weight gradient for stride 1 (maed_conv3x3_wgrad: the TN weight-gradient kernel over gathered
    rows); the three stride-2 convolutions keep the framework's convolution backward.
    w: the standardised weight as WeightStdFn hands it out, logical (O, I, 3, 3) over (O, 3, 3, I) storage.
    wt / dw (optional, both from WeightStdFn like Conv1x1Fn's): the transposed image (3,3,I,O) -- the input gradient then reads it in
    place (no flipped copy) -- and the fp32 (O, 9*I) slice the weight gradient accumulates into (autograd then carries no dW)."""

    @staticmethod
    def forward(ctx, x, w, stride, wt=None, dw=None, gn_sums=None, prec=None):
        x32, w32 = shadow_of(x), shadow_of(w)   # bf16 graph over fp32 shadows: see Conv1x1Fn
        assert (x32 is None) == (w32 is None), "Conv3x3Fn: activation and weight must both (or neither) have fp32 shadows"
        x = x.contiguous(memory_format=torch.channels_last)
        ctx.save_for_backward(x, w)
        ctx.stride, ctx.wt, ctx.dw, ctx.prec = stride, wt, dw, prec
        w_taps = (w if w32 is None else w32).permute(0, 2, 3, 1)
        w_taps = w_taps if w_taps.is_contiguous() else w_taps.contiguous()
        if x32 is not None:
            return twin_later(conv3x3(x32, w_taps, stride, gn_sums=gn_sums, prec=prec))
        return conv3x3(x, w_taps, stride, gn_sums=gn_sums, prec=prec)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        s, wt, dw_slice, prec = ctx.stride, ctx.wt, ctx.dw, ctx.prec
        dy = dy.contiguous(memory_format=torch.channels_last)
        N, I, H, W = x.shape
        O = w.shape[0]
        Ho, Wo = dy.shape[-2:]
        ph, pw = max((Ho - 1) * s + 3 - H, 0), max((Wo - 1) * s + 3 - W, 0)
        need_x = ctx.needs_input_grad[0]
        need_w = ctx.needs_input_grad[1] or dw_slice is not None
        dx = dw = None
        own_dx = need_x and s == 1 and O % 64 == 0               # (the gathered operand's channel count is O here)
        own_dx_s2 = need_x and s == 2 and wt is not None and O % 64 == 0 and I % 8 == 0 and dy.dtype == torch.bfloat16 and min(H, W) > 1
        if own_dx_s2:                                            # one implicit GEMM per parity class of the input pixel, transposed image read in place
            dx = conv3x3_s2_dgrad(dy, wt, H, W, ph // 2, pw // 2)
            own_dx = True
        elif own_dx:
            if wt is not None:                                   # in place from the transposed image: tap flip = negative tap stride
                dx = conv3x3(dy, wt, 1, w_layout=1, prec=bwd_prec(prec))
            else:                                                # dX = conv3x3(dY, w'), w'[ci][ky][kx][co] = w[co][ci][2-ky][2-kx]
                dx = conv3x3(dy, w.flip(2, 3).permute(1, 2, 3, 0).contiguous(), 1, prec=bwd_prec(prec))
        own_dw = need_w and s == 1 and ((N * H * W) % 64 == 0 or x.dtype == torch.float32) and I % 8 == 0 and O % 8 == 0      # (the bf16 kernel has no ragged tile)
        own_dw_s2 = (need_w and s == 2 and dw_slice is not None and x.dtype == torch.bfloat16 and (N * Ho * Wo) % 64 == 0 and I % 8 == 0 and O % 8 == 0
                     and N * H * W < 1 << 31)
        if own_dw_s2:                                            # the TN kernel over rows gathered through per-output-pixel tables (stride 2)
            side_stream_run(lambda: conv3x3_s2_wgrad(dy, x, ph // 2, pw // 2, out=dw_slice), dy, x, dw_slice)
            need_w = False
        elif own_dw:                                             # TN GEMM over gathered rows; fp32, (O,3,3,I) like w's storage
            if dw_slice is not None:
                side_stream_run(lambda: conv3x3_wgrad(dy, x, out=dw_slice, prec=bwd_prec(prec)), dy, x, dw_slice)
            else:
                dw = conv3x3_wgrad(dy, x, prec=bwd_prec(prec)).permute(0, 3, 1, 2)
            need_w = False
        if need_w or (need_x and not own_dx):
            sym = ph % 2 == 0 and pw % 2 == 0
            xin = x if sym else torch.nn.functional.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])
            pad = (ph // 2, pw // 2) if sym else (0, 0)
            gx, gw, _ = torch.ops.aten.convolution_backward(dy, xin, w, None, (s, s), pad, (1, 1), False, (0, 0), 1,
                                                            (need_x and not own_dx, need_w, False))
            if need_w:
                if dw_slice is not None:
                    dw_slice.view(O, 3, 3, I).add_(gw.permute(0, 2, 3, 1))
                else:
                    dw = gw
            if need_x and not own_dx:
                dx = gx if sym else gx[:, :, ph // 2:ph // 2 + H, pw // 2:pw // 2 + W]
        return dx, dw, None, None, None, None, None
