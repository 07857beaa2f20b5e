#!/usr/bin/env python3
"""Where the box-constrained headline step spends its time: line-search passes (max_linesearch_iter 1 vs 10)
and pnqp trips (pnqp_iter 1 vs 20)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import bench
from mpc import _native
from mpc._native import StepOptions
from tools.bench_extra import timed
be = _native.HipBackend()
p = bench.make_problem(12, 4, 50, 4096, torch.float32, "cuda:0", seed=5, u_scale=0.3, clamp=1.0)
out = {}
for name, o in (("ls10_qp20", StepOptions(u_lower=-1.0, u_upper=1.0)),
                ("ls1_qp20", StepOptions(u_lower=-1.0, u_upper=1.0, max_linesearch_iter=1)),
                ("ls1_qp1", StepOptions(u_lower=-1.0, u_upper=1.0, max_linesearch_iter=1, pnqp_iter=1)),
                ("ls1_qp2", StepOptions(u_lower=-1.0, u_upper=1.0, max_linesearch_iter=1, pnqp_iter=2)),
                ("wide_bounds_ls10", StepOptions(u_lower=-100.0, u_upper=100.0))):
    r = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], o)
    out[name] = {"us": round(1e3 * timed(lambda: be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], o), n=10), 1),
                 "mean_qp_iters": float(r["qp_iters"].float().mean()), "alpha_lt1": float((r["alphas"] < 1).float().mean())}
print(json.dumps(out, indent=1))
