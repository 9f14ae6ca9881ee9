"""GPU parity of the SURVEY 8(f) rows beyond the benchmarked default path -- the other st_modes, the iterative decoder, the evaluation
kernels, long-sequence attention and the implicit-GEMM 3x3 convolution -- against fixtures produced by the reference's own code
(g12, g13, g14) and the fp64 oracle: the checks of scripts/check_new_paths.py as pytest cases.  First green on MI355X in round 2
(profiles/r02_call1_new_paths_on_gpu.log)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu]


def _checks():
    spec = importlib.util.spec_from_file_location("check_new_paths", os.path.join(ROOT, "scripts", "check_new_paths.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("part", ["conv3x3", "long_attention", "modes", "iterative", "evaluation"])
def test_new_path_parity_on_gpu(part):
    """every comparison of the part runs (a mismatch does not hide the ones behind it), each goes into the parity report, and the failure message names
    exactly the comparisons that are out of tolerance"""
    from _util import note
    mod = _checks()
    mod.RESULTS.clear()
    getattr(mod, part)()
    assert mod.RESULTS, f"{part}: no comparison ran"
    for name, err, tol, ok in mod.RESULTS:
        note(f"{'ok  ' if ok else 'FAIL'} [{part}] {name:70s} rel err {err:.3e} (tol {tol:.0e})")
    bad = [f"{name}: rel err {err:.3e} > {tol:.0e}" for name, err, tol, ok in mod.RESULTS if not ok]
    assert not bad, f"{part}: {len(bad)} of {len(mod.RESULTS)} comparisons out of tolerance:\n  " + "\n  ".join(bad)
