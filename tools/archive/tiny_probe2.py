#!/usr/bin/env python3
"""Lane-per-problem kernel on the iLQR solves of configs 2 / 3: what the sweep (with the simulator's Jacobian inside), one rollout
pass and the replay cost, at the nominal of iteration `IT` of the real solve."""
import json, os, sys, copy, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
from mpc import _native, mpc
from mpc.mpc import QuadCost
from mpc._native import StepOptions
from tools.bench_extra import timed
from tools.bench_ilqr_env import problem
be = _native.HipBackend()
out = {}
for kind, B, T in (("pendulum", 1024, 20), ("cartpole", 4096, 25)):
    dx, plain, x0, Q, pp = problem(kind, B, T)
    for IT in (1, 5):
        ctrl = mpc.MPC(dx.n_state, 1, T, u_lower=dx.lower, u_upper=dx.upper, lqr_iter=IT, verbose=-1, exit_unconverged=False,
                       detach_unconverged=False, linesearch_decay=dx.linesearch_decay, max_linesearch_iter=dx.max_linesearch_iter,
                       grad_method=mpc.GradMethods.AUTO_DIFF, eps=1e-12, backprop=False, not_improved_lim=10 ** 6)
        x, u, _ = ctrl(x0, QuadCost(Q, pp), dx)
        x, u = x.detach().contiguous(), u.detach().contiguous()
        env = dx.native_env()
        env.linearize = True
        res = {}
        for name, ls, sweep in (("sweep+1pass", 1, False), ("real", dx.max_linesearch_iter, False)):
            o = StepOptions(u_lower=dx.lower, u_upper=dx.upper, linesearch_decay=dx.linesearch_decay, max_linesearch_iter=ls,
                            true_dynamics=env, sweep_only=sweep)
            for impl in (4, 6):
                plan = be.plan_step(x0, Q, pp, None, None, x, u, o, impl=impl)
                r = plan(); torch.cuda.synchronize()
                res[name + "_impl%d" % impl] = round(1e3 * timed(plan, n=20), 1)
            if name == "real":
                res["alpha_lt1"] = float((r["alphas"] < 1).float().mean())
        out["%s_it%d" % (kind, IT)] = res
print(json.dumps(out, indent=0))
