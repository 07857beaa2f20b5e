#!/usr/bin/env python3
"""What a trip of the box QP costs INSIDE the box-constrained headline step (round 4): the same launch with pnqp's iteration
cap at 1, 2, 3, 4, 20 (mpc/pnqp.py:5 n_iter; the results of the capped runs are not the reference's -- timing only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import torch, bench
from mpc import _native
from mpc._native import StepOptions
be = _native.HipBackend()
for ns, nc, T, B in ((12, 4, 50, 4096), (32, 8, 64, 1024)):
    p = bench.make_problem(ns, nc, T, B, torch.float32, "cuda:0", seed=5, u_scale=0.3, clamp=1.0, on_device=ns > 16)
    a = (p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"])
    for it in (1, 2, 3, 4, 20):
        plan = be.plan_step(*a, StepOptions(u_lower=-1.0, u_upper=1.0, nominal_on_dynamics=True, c_symmetric=True, pnqp_iter=it))
        for _ in range(150):
            plan()
        _, ms, r = bench.timed(plan, 40, 0)
        print("%d/%d pnqp_iter <= %2d: %.4f ms   (mean QP iterations per problem-step %.2f)" % (ns, nc, it, ms, float(r["qp_iters"].float().mean()) / T))
