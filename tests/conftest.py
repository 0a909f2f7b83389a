import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present():
    try:
        from mppi_gpu_amd import _capi
        return _capi.load().mppi_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests FAIL (not skip) when the HIP library or the device is missing: a silent skip
    would hide a product path that does not run."""
    from mppi_gpu_amd import _capi
    lib = _capi.load()
    assert lib.mppi_device_count() > 0, "no HIP device visible to libmppi_gpu_amd.so"
    return lib


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
