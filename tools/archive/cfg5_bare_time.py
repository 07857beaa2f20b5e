import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import torch, bench
from mpc import _native
from mpc._native import StepOptions
be = _native.HipBackend()
for B in (1024, 8192):
    p = bench.make_problem(32, 8, 64, B, torch.float32, "cuda:0", seed=9)
    a = (p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"])
    for name, o in (("vouched+sym", StepOptions(nominal_on_dynamics=True, c_symmetric=True)), ("sym only (nominal verified in the sweep)", StepOptions(c_symmetric=True)),
                    ("bare (nominal and C verified)", StepOptions())):
        plan = be.plan_step(*a, o)
        _, ms, _ = bench.timed(plan, 30, 10)
        print("B", B, name, "us", round(ms * 1e3, 1))
