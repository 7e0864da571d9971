import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_sessionstart(session):
    """The shared library is built in-tree and git-ignored: a fresh checkout has none.  Build it once (hipcc
    cross-compiles gfx950 without a GPU, ~20 s) so that the suite tests the sources it was checked out with;
    without hipcc the tests that load it fail with the loader's own message (there is no CPU fallback)."""
    import shutil
    import subprocess
    so = os.path.join(ROOT, "gpry_amd", "libgpry_hip.so")
    if os.path.exists(so) or os.environ.get("GPRY_HIP_LIB"):
        return
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "gpry_amd", "csrc")], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden
