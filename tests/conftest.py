import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def backend():
    """The HIP backend; GPU tests FAIL (not skip) when the library or the device is missing."""
    from glimslib_amd import _backend
    _backend.load_library()
    return _backend
