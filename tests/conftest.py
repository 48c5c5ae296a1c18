import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")

# torch ships its own HIP runtime (libamdhip64 inside the wheel); libvoicebridge_hip.so names the same SONAME and gets whichever copy
# the process loaded first.  A test that loads the host library through ctypes before anything imported torch would bring in the system
# copy, and torch's own copy, loaded later, then finds no device ("No HIP GPUs are available").  Importing torch first makes every
# order of the test files use one runtime.
import torch  # noqa: E402,F401
