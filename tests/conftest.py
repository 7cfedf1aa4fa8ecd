import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def hip():
    """The product's C-ABI binding.  No fallback: a missing library is an error."""
    from popsift_amd import _capi
    _capi.lib()
    return _capi


@pytest.fixture(scope="session")
def gpu_hip(hip):
    if hip.device_count() < 1:
        pytest.fail("GPU test selected but libpopsift_hip sees no device (native path must run here)")
    return hip
