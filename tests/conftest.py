import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """The built C-ABI library (built on demand with hipcc; cross-compiles without a GPU)."""
    from se_snmf_nat_amd import _lib
    _lib.build()
    return _lib.load()


@pytest.fixture(scope="session")
def gpu_ctx(lib):
    from se_snmf_nat_amd import Context
    if lib.snmf_device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible (no CPU fallback exists)")
    return Context(0)
