#!/usr/bin/env python3
"""What the symmetry test of C costs at the headline shape (round 3): interleaved timing, one box, of
  promised      MPC_OPT_NOMINAL_ON_DYNAMICS | MPC_OPT_C_SYMMETRIC   (a steady-state mpc.MPC iteration)
  test_only     nominal vouched, C not: impl 3 forced -> the kernel's column reads, no second launch
  test_gate     nominal vouched, C not: impl 0 -> + the gated launch of the generic kernel
  bare          no promise at all (a bare LQRStep call)
  bounded_*     the same two ends for the box-constrained step
usage: python tools/ab_sym.py [reps]      (MPC_LQR_HIP_LIB selects a library variant)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from mpc import _native
from mpc._native import StepOptions

be = _native.HipBackend()
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
p = bench.make_problem(12, 4, 50, 4096, torch.float32, dev, seed=1000)
pb = bench.make_problem(12, 4, 50, 4096, torch.float32, dev, seed=1000, u_scale=0.3, clamp=1.0)
bd = dict(u_lower=-1.0, u_upper=1.0)
cases = {
    "promised": (p, StepOptions(nominal_on_dynamics=True, c_symmetric=True), 0),
    "test_only": (p, StepOptions(nominal_on_dynamics=True), 3),
    "test_gate": (p, StepOptions(nominal_on_dynamics=True), 0),
    "bare": (p, StepOptions(), 0),
    "bounded_promised": (pb, StepOptions(nominal_on_dynamics=True, c_symmetric=True, **bd), 0),
    "bounded_bare": (pb, StepOptions(**bd), 0),
}
plans = {k: be.plan_step(q["x_init"], q["C"], q["c"], q["F"], q["f"], q["cur_x"], q["cur_u"], o, impl=i) for k, (q, o, i) in cases.items()}
for _ in range(150):
    plans["promised"]()
res = {k: [] for k in plans}
for r in range(reps):
    for k, pl in plans.items():
        _, ms, _ = bench.timed(pl, 40, 10)
        res[k].append(round(ms * 1e3, 2))
print(json.dumps({"lib": os.environ.get("MPC_LQR_HIP_LIB", "default"), "us_per_launch": res}))
