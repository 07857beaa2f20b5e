import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mpc.pytorch_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# the library reads MPC_DPP16_RING (which staging ring the 12/4 kernel runs on) once, when it is loaded; with this set it
# follows the environment from launch to launch, so a test can hold both rings to the oracle in one process
os.environ.setdefault("MPC_DPP16_RING_DYNAMIC", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    import numpy as np
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture
def load_golden():
    return golden


@pytest.fixture(scope="session")
def emu_libs():
    """tests/emu_backend.py's four emulator libraries, built in parallel before the first test that runs a kernel body on the CPU."""
    import shutil
    if not (os.path.exists("/opt/rocm/lib/llvm/bin/clang++") or shutil.which("clang++")):
        pytest.skip("the emulator needs clang++")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_backend
    emu_backend.build_all()
    return emu_backend

