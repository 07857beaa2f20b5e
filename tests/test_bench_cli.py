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
