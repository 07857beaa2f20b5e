#!/usr/bin/env python3
"""Lane-per-problem (impl 4) against row-per-problem (impl 6) kernel over the batch size, at the nominal of iteration 5 of the
real iLQR solve of configs 2 / 3:  python tools/w1_bsweep.py"""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
from mpc import _native, mpc
from mpc.mpc import QuadCost
from mpc._native import StepOptions
from tools.bench_extra import timed
from tools.bench_ilqr_env import problem
be = _native.HipBackend()
out = {}
for kind, T in (("pendulum", 20), ("cartpole", 25)):
    for B in (256, 512, 1024, 2048, 3072, 4096, 8192, 16384, 32768):
        dx, plain, x0, Q, pp = problem(kind, B, T)
        ctrl = mpc.MPC(dx.n_state, 1, T, u_lower=dx.lower, u_upper=dx.upper, lqr_iter=5, verbose=-1, exit_unconverged=False,
                       detach_unconverged=False, linesearch_decay=dx.linesearch_decay, max_linesearch_iter=dx.max_linesearch_iter,
                       grad_method=mpc.GradMethods.AUTO_DIFF, eps=1e-12, backprop=False, not_improved_lim=10 ** 6)
        x, u, _ = ctrl(x0, QuadCost(Q, pp), dx)
        x, u = x.detach().contiguous(), u.detach().contiguous()
        env = dx.native_env()
        env.linearize = True
        o = StepOptions(u_lower=dx.lower, u_upper=dx.upper, linesearch_decay=dx.linesearch_decay, max_linesearch_iter=dx.max_linesearch_iter,
                        true_dynamics=env)
        res = {}
        for impl in (4, 6):
            plan = be.plan_step(x0, Q, pp, None, None, x, u, o, impl=impl)
            plan(); torch.cuda.synchronize()
            res["impl%d" % impl] = round(1e3 * timed(plan, n=20), 1)
        out["%s_B%d" % (kind, B)] = res
        print(kind, B, res, flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "w1_bsweep.json"), "w"), indent=1)
