import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.load()
    return oracle_lib


@pytest.fixture(scope="session")
def product_lib():
    """Builds (if stale) and loads librayca_hip.so.  No fallback: a missing library is an error."""
    import __graft_entry__ as g
    g.build()
    from rayca_amd import lib
    return lib.load()


@pytest.fixture(scope="session")
def gpu(product_lib):
    if product_lib.rayca_hip_device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible (there is no CPU fallback)")
    return product_lib
