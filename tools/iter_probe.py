#!/usr/bin/env python3
"""Round 5: what the LATER iterations of a box-constrained solve cost and why (mpc_forward_5iter_bounded).
Five LQR steps in a row at the headline shape, each from the previous one's result: kernel time, the distribution of the accepted
step sizes, pnqp iterations per timestep.   python tools/r05_iter_probe.py [iters]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import bench
from mpc import _native
from mpc._native import StepOptions
be = _native.HipBackend()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
p = bench.make_problem(12, 4, 50, 4096, torch.float32, "cuda:0", seed=5, u_scale=0.3, clamp=1.0)
opts = StepOptions(u_lower=-1.0, u_upper=1.0, nominal_on_dynamics=True, c_symmetric=True)
x, u = p["cur_x"], p["cur_u"]
for i in range(iters):
    plan = be.plan_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], x, u, opts)
    for _ in range(30):
        plan()
    _, ms, r = bench.timed(plan, 20, 0)
    al = r["alphas"].cpu().numpy()
    vals, cnt = np.unique(np.round(np.log(al) / np.log(0.2)).astype(int), return_counts=True)
    dc = (r["costs"] - r["old_costs"]).cpu().numpy()
    print("iteration %d: %.1f us  qp/t %.2f  alpha = 0.2^k counts %s  cost change: median %.3g, share > 0: %.3f, |du| max %.3g  mean cost %.6g" % (
        i, ms * 1e3, float(r["qp_iters"].float().mean()) / 50, dict(zip(vals.tolist(), cnt.tolist())), np.median(dc), (dc > 0).mean(),
        float(r["full_du_norm"].max()), float(r["costs"].mean())))
    x, u = r["new_x"].clone(), r["new_u"].clone()
