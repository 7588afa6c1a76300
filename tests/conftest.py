import os, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# On a GPU box initialise torch's HIP runtime BEFORE libharc_amd.so is loaded: torch bundles its own libamdhip64 and refuses to
# see the device when another copy of the runtime was initialised first in the same process.
try:
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
except Exception:  # pragma: no cover
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """ctypes handle on oracle/liboracle.so (the CPU checker); built on demand with gcc."""
    from tests import oracle_lib
    return oracle_lib.load()
