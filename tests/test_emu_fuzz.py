"""A slice of tools/emu_fuzz.py in the CPU suite: the fused kernels' bodies on the wavefront emulator against the float64 oracle over
random option sets (ragged batches, horizons, bounds of every kind, delta_u, u_zero_I, promises, qp_start, line-search depth).
The tool runs thousands of cases in minutes; here a fixed seed's first cases, so that a change to a body that breaks an option
combination no parametrised test names shows up without the GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("kernels,cases,long_t", [("dpp16,dpp16_ring2,mfma16", 150, False), ("mfma40,mfma40_ring2", 24, False),
                                                   ("dpp16,dpp16_ring2", 24, True),
                                                   ("mfma16_f64", 60, False), ("mfma40_pad", 20, False), ("dpp16_pad", 80, False), ("dpp16_pad", 16, True)])
def test_emulated_bodies_on_random_option_sets(emu_libs, kernels, cases, long_t):
    if not (os.path.exists("/opt/rocm/lib/llvm/bin/clang++") or __import__("shutil").which("clang++")):
        pytest.skip("the emulator needs clang++")
    env = dict(os.environ)
    if long_t:
        env["FUZZ_LONG_T"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "emu_fuzz.py"), str(cases), "11", kernels],
                       capture_output=True, text=True, env=env, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "violations 0" in p.stdout


@pytest.mark.parametrize("which,cases", [("dpp16", 40), ("dpp16_pad", 40), ("mfma40", 16), ("mfma40_pad", 12)])
def test_emulated_kkt_backward_on_random_option_sets(emu_libs, which, cases):
    """tools/emu_fuzz_kkt.py: the fused KKT backward bodies against LQRStepFn.backward of the oracle -- horizons across the 64-step
    limit of the register-resident gains, ragged batches, bounds of every kind, f on / off, both ring builds; dpp16_pad: random shapes
    up to 12/4 through the padded instantiation (round 6)."""
    if not (os.path.exists("/opt/rocm/lib/llvm/bin/clang++") or __import__("shutil").which("clang++")):
        pytest.skip("the emulator needs clang++")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "emu_fuzz_kkt.py"), str(cases), "11", which],
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "violations 0" in p.stdout
