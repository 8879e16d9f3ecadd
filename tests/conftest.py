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


def _host_logic_knobs():
    """On a machine without a GPU the host-logic tests of this suite (CLI, file formats, the smoothing rules) drive the
    binary through its developer switches, explicitly: `SVDSS index` / `SVDSS smooth` themselves refuse to run without a
    GPU.  On the GPU box nothing is set and the HIP paths run."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:   # noqa: BLE001
        has_gpu = False
    if not has_gpu:
        os.environ.setdefault("SVDSS_INDEX_CPU", "1")
        os.environ.setdefault("SVDSS_SMOOTH_HOST", "1")


_host_logic_knobs()


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    return oracle_lib


@pytest.fixture(scope="session")
def emulator():
    from tests import emulator_lib
    return emulator_lib
