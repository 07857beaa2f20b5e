"""bench.py's launcher contract, checked without a GPU: `--gpus N` either runs N ranks or refuses loudly."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="needs a box with fewer than 2 GPUs")
def test_gpus_n_without_enough_devices_refuses_instead_of_running_one_rank():
    """VERDICT r01 / ADVICE: `python bench.py --gpus 2` used to run ONE rank and print n_gpus: 1."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "refusing" in r.stderr
    assert r.stdout.strip() == ""            # no JSON line that could be mistaken for a 2-GPU result


def test_world_size_mismatch_refuses():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "refusing" in r.stderr and r.stdout.strip() == ""


def test_byte_and_flop_formulas():
    """100,840 B / problem and 195,264 B / problem (KKT) at the headline shape; 17.3 / 256 kFLOP."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.algorithmic_bytes_per_problem(12, 4, 50) == 100840
    assert bench.kkt_algorithmic_bytes_per_problem(12, 4, 50) == 195264
    assert abs(bench.algorithmic_flops_per_problem_step(12, 4) - 17290) < 30
    assert abs(bench.algorithmic_flops_per_problem_step(32, 8) - 255862) < 400


def test_parity_object_accepts_the_oracles_own_numbers_and_rejects_a_wrong_trajectory():
    """bench.py's in-run self-certification: `parity.ok` for results within rtol 1e-3 /
    atol 1e-4 of the oracle on the first 64 problems of the timed batch, not ok (and the run exits non-zero) otherwise."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    from oracle import lqr_oracle as O
    rng = np.random.default_rng(3)
    T, B, ns, nc = 12, 70, 12, 4
    n = ns + nc
    A = rng.standard_normal((T, B, n, n)).astype(np.float32)
    C = np.einsum("tbji,tbjk->tbik", A, A)
    c = rng.standard_normal((T, B, n)).astype(np.float32)
    F = np.concatenate((np.eye(ns) + 0.2 * rng.standard_normal((T - 1, B, ns, ns)) / np.sqrt(ns),
                        rng.standard_normal((T - 1, B, ns, nc)) / np.sqrt(ns)), 3).astype(np.float32)
    f = (0.1 * rng.standard_normal((T - 1, B, ns))).astype(np.float32)
    x0 = rng.standard_normal((B, ns)).astype(np.float32)
    u = np.zeros((T, B, nc), np.float32)
    x, _ = O.traj_cost(x0.astype(np.float64), u.astype(np.float64), F.astype(np.float64), f.astype(np.float64))
    p = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in dict(x_init=x0, C=C, c=c, F=F, f=f, cur_x=x.astype(np.float32), cur_u=u).items()}
    o = O.lqr_step(*(p[k].numpy().astype(np.float64) for k in ("x_init", "C", "c", "F", "f", "cur_x", "cur_u")), None, None, lockstep=False)
    r = {k: torch.from_numpy(o[k].astype(np.float32)) for k in ("new_x", "new_u", "costs", "alphas")}
    good = bench.parity_check(p, r, False)
    assert good["ok"] and good["problems"] == 64 and good["max_err_over_tol_u"] < 0.1
    r["new_u"][3, 5, 1] += 0.01
    bad = bench.parity_check(p, r, False)
    assert not bad["ok"] and bad["max_err_over_tol_u"] > 1.0
