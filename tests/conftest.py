import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """Build the product library and the oracle if they are missing (CPU-only; hipcc cross-compiles)."""
    lib = os.path.join(ROOT, "svdss_amd", "libsvdss_hip.so")
    orc = os.path.join(ROOT, "oracle", "libsvdss_oracle.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "svdss_amd", "csrc")])
    if not os.path.exists(orc):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


_ensure_built()


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    return oracle_lib


@pytest.fixture(scope="session")
def emulator():
    from tests import emulator_lib
    return emulator_lib
