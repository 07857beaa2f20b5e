#!/usr/bin/env python3
"""Forward + backward through MPC for the shipped simulators (configs 2 / 3 as a differentiable layer: gradients w.r.t. the cost)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
from mpc import mpc
from mpc.mpc import QuadCost
from tools.bench_ilqr_env import problem
for kind, B, T in (("pendulum", 1024, 20), ("cartpole", 4096, 25)):
    dx, _plain, x0, Q, pp = problem(kind, B, T)
    Q = Q.clone().requires_grad_(True); pp = pp.clone().requires_grad_(True)
    ctrl = mpc.MPC(dx.n_state, 1, T, u_lower=dx.lower, u_upper=dx.upper, lqr_iter=10, verbose=-1, exit_unconverged=False,
                   detach_unconverged=False, linesearch_decay=dx.linesearch_decay, max_linesearch_iter=dx.max_linesearch_iter,
                   grad_method=mpc.GradMethods.AUTO_DIFF, eps=1e-12, backprop=True, not_improved_lim=10 ** 6)
    def once():
        x, u, c = ctrl(x0, QuadCost(Q, pp), dx)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        (x.sum() + u.sum()).backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        Q.grad = None; pp.grad = None
        return t2 - t1
    for _ in range(3): once()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bw = 0.0
    for _ in range(10): bw += once()
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / 10
    print(kind, "forward+backward ms", round(tot * 1e3, 3), "of which backward ms", round(bw / 10 * 1e3, 3), flush=True)
