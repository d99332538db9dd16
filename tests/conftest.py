import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    d = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (d if d > 0 else 1.0)


def maxrel(a, b):
    """max |a-b| / max|b| with NaN patterns required identical."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert np.array_equal(np.isnan(a), np.isnan(b)), "NaN masks differ"
    m = ~np.isnan(b)
    if not m.any():
        return 0.0
    s = np.max(np.abs(b[m]))
    return np.max(np.abs(a[m] - b[m])) / (s if s > 0 else 1.0)


def pointrel(a, b):
    """max over the points of |a-b| / |b| (NaN patterns identical): for fields that span decades (GEOM-mean viscosities), where
    a tolerance relative to the global maximum says little at the low-viscosity nodes."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert np.array_equal(np.isnan(a), np.isnan(b)), "NaN masks differ"
    m = ~np.isnan(b) & (b != 0)
    if not m.any():
        return 0.0
    return float(np.max(np.abs(a[m] - b[m]) / np.abs(b[m])))


@pytest.fixture(scope="session")
def oracle():
    from oracle import pylamp_oracle
    return pylamp_oracle
