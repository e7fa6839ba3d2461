import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionstart(session):
    """The CPU suite checks that libpfdyn.so loads and exports the declared ABI, so build it when it is missing
    (hipcc cross-compiles gfx950 without a GPU).  GPU boxes receive the prebuilt .so with the snapshot."""
    lib = os.path.join(ROOT, "pharmacophore-diffusion_amd", "csrc", "libpfdyn.so")
    if not os.path.exists(lib):
        import shutil
        import subprocess
        if shutil.which("hipcc") and shutil.which("make"):
            subprocess.run(["make", "-C", os.path.dirname(lib), "-j4"], check=False)
