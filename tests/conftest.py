import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    path = os.path.join(GOLDEN, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"golden fixture {name}.npz not generated")
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def relmax(a, b):
    """max|a-b| / max|b| — the array-level parity metric of SURVEY §7 'hard parts'."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


def assert_parity(a, b, rtol=1e-5, what=""):
    """north_star tolerance: 1e-5 relative, fp64.  Array-level max-norm relative
    error plus an elementwise rtol/atol (atol = rtol * array scale) check."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    assert np.all(np.isfinite(a)), f"{what}: non-finite values"
    scale = float(np.max(np.abs(b))) or 1.0
    err = relmax(a, b)
    assert err <= rtol, f"{what}: max-norm relative error {err:.3e} > {rtol:g}"
    np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * scale, err_msg=what)
