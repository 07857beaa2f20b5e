"""A slice of tools/ref_diff_mpc.py in the CPU suite: mpc.MPC's host logic (iteration, best-iterate bookkeeping, convergence exits,
detach_unconverged, the autograd wiring of LQRStep) on the oracle-backed stand-in against the UNMODIFIED reference run in a child
process.  Build container only: skipped where /root/reference (or $MPC_REFERENCE_DIR) does not exist -- never part of the GPU run."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MPC_REFERENCE_DIR", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "mpc")), reason="the reference is not on this box")
def test_mpc_host_logic_against_the_unmodified_reference_on_random_configurations():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_diff_mpc.py"), "120", "17"], capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "violations 0" in p.stdout


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "mpc")), reason="the reference is not on this box")
def test_the_oracle_against_the_unmodified_reference_on_random_problems():
    """tools/ref_diff_oracle.py: the pin of tests/test_oracle_golden.py (72 fixtures) on random problems -- LQRStep forward with
    lockstep semantics and LQRStepFn.backward through the reference's own autograd, float64, 1e-7."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_diff_oracle.py"), "150", "17"], capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "violations 0" in p.stdout


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "mpc")), reason="the reference is not on this box")
def test_the_oracle_other_entry_points_against_the_unmodified_reference():
    """tools/ref_diff_misc.py: pnqp (solution, free set, iteration count), get_traj / get_cost, NNDynamics forward and grad_input."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_diff_misc.py"), "240", "17"], capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "violations 0" in p.stdout
