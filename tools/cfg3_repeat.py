#!/usr/bin/env python3
"""Config 3 (cart-pole iLQR, B = 4096, T = 25, 10 iterations) timed the way bench.py times it, several times in one process:
is the solve time stable?  (bench.py rows of 0.70 and 1.8 ms have both been seen.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import torch, bench
from mpc import mpc
from mpc.mpc import QuadCost
from tools.bench_ilqr_env import problem as env_problem
if "--pre" in sys.argv:
    # what bench.py's config-5 rows leave behind: large blocks allocated, used and handed back to the driver
    ps = [bench.make_problem(32, 8, 64, 8192, torch.float32, "cuda:0", seed=9, on_device=True)]
    big = torch.zeros(800 * 1024 * 1024, dtype=torch.uint8, device="cuda:0")
    view = big[::4096]
    for _ in range(40):
        view.sum()
    torch.cuda.synchronize()
    del ps, big, view
    if "--keep" not in sys.argv:
        torch.cuda.empty_cache()
kind, B, T = ("pendulum", 1024, 20) if "--pendulum" in sys.argv else ("cartpole", 4096, 25)
dxm, _plain, x0, Q, pp = env_problem(kind, B, T)
ctrl = mpc.MPC(dxm.n_state, 1, T, u_lower=dxm.lower, u_upper=dxm.upper, lqr_iter=10, verbose=-1,
               exit_unconverged=False, detach_unconverged=False, linesearch_decay=dxm.linesearch_decay,
               max_linesearch_iter=dxm.max_linesearch_iter, grad_method=mpc.GradMethods.AUTO_DIFF,
               eps=1e-12, backprop=False, not_improved_lim=10 ** 6)
cost = QuadCost(Q, pp)
for rep in range(6):
    wall, ms, out = bench.timed(lambda: ctrl(x0, cost, dxm), 5, 2)
    # one more solve, timed on the host alone, with the longest single wait of the flag reader
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ctrl(x0, cost, dxm); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("rep", rep, "ms_per_solve(events)", round(ms, 3), "wall", round(wall, 3), "single solve wall ms", round((t1 - t0) * 1e3, 3), flush=True)
