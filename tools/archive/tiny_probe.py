#!/usr/bin/env python3
"""Lane-per-problem kernel: share of the line-search passes (pendulum / cart-pole step, simulator rollout)."""
import json, os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
from mpc import _native
from mpc._native import StepOptions
from mpc.env_dx import pendulum, cartpole
from tools.bench_extra import timed
from tools.bench_ilqr_env import problem
be = _native.HipBackend()
out = {}
for kind, B, T in (("pendulum", 1024, 20), ("cartpole", 4096, 25)):
    dx, plain, x0, Q, pp = problem(kind, B, T)
    env = dx.native_env()
    u = torch.zeros(T, B, 1, device="cuda:0")
    x, _ = be.env_traj_cost(x0, u, env)
    F, f = be.env_linearize(env, x[:-1].reshape(-1, dx.n_state), u[:-1].reshape(-1, 1))
    F, f = F.view(T - 1, B, dx.n_state, -1), f.view(T - 1, B, -1)
    for ls in (1, dx.max_linesearch_iter, 10):
        o = StepOptions(u_lower=dx.lower, u_upper=dx.upper, linesearch_decay=dx.linesearch_decay, max_linesearch_iter=ls, true_dynamics=env)
        plan = be.plan_step(x0, Q, pp, F, f, x, u, o)
        r = plan(); torch.cuda.synchronize()
        out["%s_ls%d" % (kind, ls)] = {"us": round(1e3 * timed(plan, n=20), 1), "alpha_lt1": float((r["alphas"] < 1).float().mean()),
                                       "alpha_min": float(r["alphas"].min())}
print(json.dumps(out))
