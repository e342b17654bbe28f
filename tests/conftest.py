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


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]
    return load


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


def assert_close(a, b, rtol=1e-12, atol_scale=1e-13, msg=""):
    """|a - b| <= rtol |b| + atol_scale * max|b| element-wise (finite entries of b only)."""
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    assert a.shape == b.shape, "%s shape %s vs %s" % (msg, a.shape, b.shape)
    fin = np.isfinite(b)
    scale = np.abs(b[fin]).max() if fin.any() else 1.0
    np.testing.assert_allclose(a[fin], b[fin], rtol=rtol, atol=atol_scale * scale, err_msg=msg)


def assert_close_nan(a, b, rtol=1e-12, atol_scale=1e-13, msg=""):
    """assert_close for results that may hold NaN / +-inf by design (MaternKernel at tau == 0): non-finite entries must
    agree in kind and sign, finite ones within rtol / atol_scale."""
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    assert a.shape == b.shape, "%s shape %s vs %s" % (msg, a.shape, b.shape)
    np.testing.assert_array_equal(np.isnan(a), np.isnan(b), err_msg=msg + " (NaN pattern)")
    inf = np.isinf(b)
    np.testing.assert_array_equal(np.isinf(a), inf, err_msg=msg + " (inf pattern)")
    np.testing.assert_array_equal(np.sign(a[inf]), np.sign(b[inf]), err_msg=msg + " (inf sign)")
    fin = np.isfinite(b)
    scale = np.abs(b[fin]).max() if fin.any() else 1.0
    np.testing.assert_allclose(a[fin], b[fin], rtol=rtol, atol=atol_scale * scale, err_msg=msg)
