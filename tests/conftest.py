"""pytest configuration.

  -m "not gpu" : oracle vs golden vectors / vs the real reference (when
                 oracle/_ref exists), host logic, C-ABI load + symbol check.
  -m gpu       : parity tests proper — the HIP path through the C ABI
                 (leansdr_amd/liblsdr_hip.so) against the oracle and the goldens.
"""
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    return pyoracle.Oracle()  # builds liblsdr_oracle.so on demand


@pytest.fixture(scope="session")
def ref():
    import pyoracle
    if not pyoracle.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    return pyoracle.Ref()


@pytest.fixture(scope="session")
def capi():
    import leansdr_amd.capi as c  # raises ImportError if liblsdr_hip.so is missing (no fallback)
    return c


@pytest.fixture(scope="session")
def ctx(capi):
    c = capi.Ctx(0)  # raises on a machine without a GPU — gpu tests must not silently pass
    yield c
    c.close()


def gold(name):
    return np.load(os.path.join(GOLD, name))


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def iq16_to_cf32(iq):
    return (iq[:, 0].astype(np.float32) + 1j * iq[:, 1].astype(np.float32)).astype(np.complex64)
