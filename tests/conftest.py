import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def hip_lib():
    import pyspeedy_amd
    if not os.path.isfile(pyspeedy_amd._lib.LIB_PATH):
        pyspeedy_amd.build()
    return pyspeedy_amd.lib()


@pytest.fixture(scope="session")
def spectral():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    import pyspeedy_amd
    return pyspeedy_amd.ModSpectral()
