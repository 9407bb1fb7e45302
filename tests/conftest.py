import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    oracle.build()
    return oracle


def _gpu_present():
    try:
        from zoic_amd import _capi
        return _capi.load().zoic_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must not pass on a fallback: the HIP library has to load and see a device."""
    from zoic_amd import _capi
    lib = _capi.load()
    n = lib.zoic_device_count()
    assert n > 0, "no HIP device visible but a gpu-marked test was selected"
    return lib
