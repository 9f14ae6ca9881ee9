"""TEST INFRASTRUCTURE: run the decoder-tail / loss kernels of libmaed_hip on the host simulator
(tests/hostsim) so their arithmetic and the ctypes/autograd wiring are checked without a GPU.
`patched()` swaps the library handle and the pointer/stream helpers for the duration of a test only."""
import contextlib
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "hostsim"))


def load():
    import build_sim
    from maed_amd import _lib as L
    h = C.CDLL(build_sim.build())
    for name, (res, args) in L.SIGNATURES.items():
        if hasattr(h, name):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
    assert h.maed_version() < 0, "this must be the simulator, not the product library"
    L.apply_options(h)          # the host's option values (fp32 matmul mode, side stream ...) as the product library would get them
    # the LBS blend on the fp32 matrix cores (round 5) costs the simulator ~150 k emulated MFMA rendezvous per call: the suites that merely pass through the decoder
    # tail run the VALU kernel (identical interface); tests/test_hostsim_tail.py::test_lbs_blend_on_the_matrix_cores_matches_the_valu_kernel covers the MFMA path
    if "MAED_LBS_FB" not in os.environ:
        h.maed_set_option(L.OPT_LBS_FRAMES, 4)
    return h


@contextlib.contextmanager
def patched(module_paths=True):
    """module_paths=False: only explicit library calls run on the simulator; module-level gates (ops.on_library_device: the backbone's
    kernel path) stay on their ATen CPU path -- for tests about something else that should not pay for ~700 simulated workgroups per
    weight-standardisation launch"""
    from maed_amd import _lib as L
    from maed_amd import ops
    saved = (L._lib, ops._p, ops._stream, ops.SIM_MODULE_PATHS)
    L._lib = load()
    ops._p = lambda t: None if t is None else t.data_ptr()
    ops._stream = lambda: None
    ops.SIM_MODULE_PATHS = module_paths
    try:
        yield L._lib
    finally:
        L._lib, ops._p, ops._stream, ops.SIM_MODULE_PATHS = saved


@contextlib.contextmanager
def option(lib, key, value):
    """one library option (maed_set_option) for the duration of a block; the simulator handle is shared between tests, so the old value is put back"""
    old = lib.maed_get_option(key)
    assert lib.maed_set_option(key, value) == 0, lib.maed_last_error()
    try:
        yield
    finally:
        lib.maed_set_option(key, old)
