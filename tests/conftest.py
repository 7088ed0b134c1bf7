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


# Tests that start other programs (the Fortran host, torch.distributed.run ranks) run FIRST: the GPU box does not allow a
# process that has initialised the GPU to exec another program (not even in a forked child), and the test process
# initialises it as soon as the first device fixture is used.
_SPAWNING_MODULES = ("test_fortran_host.py", "test_ensemble_dist.py", "test_bench_launch.py", "test_variants_spawn.py",
                     "test_sanitizers.py", "test_c_host.py", "test_calendar_host.py")


def pytest_sessionstart(session):
    """Build (make: a no-op when up to date) the CPU oracle and the HIP library BEFORE any test can initialise the GPU: the
    GPU box refuses to start programs (make, gcc, hipcc) from a process that has."""
    import oracle as orc
    orc.build()
    import pyspeedy_amd
    if not os.path.isfile(pyspeedy_amd._lib.LIB_PATH):
        pyspeedy_amd.build()


def pytest_collection_modifyitems(config, items):
    items.sort(key=lambda item: 0 if os.path.basename(str(item.fspath)) in _SPAWNING_MODULES else 1)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure); built with gcc in pytest_sessionstart."""
    import oracle as orc
    return orc


@pytest.fixture(scope="session")
def hip_lib():
    import pyspeedy_amd
    return pyspeedy_amd.lib()


@pytest.fixture(scope="session")
def spectral():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    import pyspeedy_amd
    return pyspeedy_amd.ModSpectral()
