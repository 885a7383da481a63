import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "3dscanner-graycode_amd")
for p in (PKG, os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_cases(fname):
    """npz with 'case/key' entries -> {case: {key: array}}."""
    z = np.load(os.path.join(GOLDEN, fname))
    out = {}
    for k in z.files:
        case, key = k.rsplit("/", 1)
        out.setdefault(case, {})[key] = z[k]
    return out


@pytest.fixture(scope="session")
def decode_cases():
    return load_cases("decode_cases.npz")


@pytest.fixture(scope="session")
def tri_cases():
    return load_cases("triangulate.npz")


@pytest.fixture(scope="session")
def calib():
    z = np.load(os.path.join(GOLDEN, "calib.npz"))
    return {k: z[k] for k in z.files}


def has_gpu():
    """True when a HIP device is usable through the product library (no torch involved)."""
    try:
        from scanner import _native
        return _native.device_count() > 0
    except Exception:
        return False
