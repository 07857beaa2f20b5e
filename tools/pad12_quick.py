"""The padded 12/4 kernel's step at a few shapes, timing only (variants: MPC_LQR_HIP_LIB).   python tools/pad12_quick.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd"))
import bench
from mpc import _native
from mpc._native import StepOptions
be = _native.HipBackend()
for bounded in (False, True):
    for ns, nc in ((8, 4), (10, 3), (12, 2), (12, 4)):
        p = bench.make_problem(ns, nc, 50, 4096, torch.float32, "cuda:0", seed=3, u_scale=0.3 if bounded else 0.0, clamp=1.0 if bounded else None)
        o = StepOptions(u_lower=-1.0, u_upper=1.0, nominal_on_dynamics=True, c_symmetric=True) if bounded else StepOptions(nominal_on_dynamics=True, c_symmetric=True)
        plan = be.plan_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], o, impl=8)
        for _ in range(150):
            plan()
        _, ms, _ = bench.timed(plan, 40, 0)
        print("%2d/%d %s impl 8: %.1f us" % (ns, nc, "bounded" if bounded else "unbounded", ms * 1e3), flush=True)
