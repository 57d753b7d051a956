import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (built from /root/reference in the build container)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the oracle (checker) and the HIP library (product) are built.
    Building the checker is not using it; the GPU box normally receives prebuilt files."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so", "liboracle_bkm.so"])
    if not os.path.exists(os.path.join(ROOT, "blinky_amd", "libblinkyhip.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "blinky_amd", "csrc")])
    yield
