#!/usr/bin/env python3
"""Where the HOST time of one MPC.forward goes (cProfile over 200 solves; the device runs ahead or idles -- this is about Python):
   python tools/host_profile.py [pendulum|cartpole|headline]"""
import cProfile, os, pstats, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import torch
from mpc import mpc
from mpc.mpc import QuadCost, LinDx
kind = sys.argv[1] if len(sys.argv) > 1 else "pendulum"
if kind == "headline":
    import bench
    p = bench.make_problem(12, 4, 50, 4096, torch.float32, "cuda:0", seed=5, on_device=True)
    ctrl = mpc.MPC(12, 4, 50, lqr_iter=5, verbose=-1, exit_unconverged=False, detach_unconverged=False, backprop=False)
    cost, dx, x0 = QuadCost(p["C"], p["c"]), LinDx(p["F"], p["f"]), p["x_init"]
else:
    from tools.bench_ilqr_env import problem
    B, T = (1024, 20) if kind == "pendulum" else (4096, 25)
    dx, _plain, x0, Q, pp = problem(kind, B, T)
    ctrl = mpc.MPC(dx.n_state, 1, T, u_lower=dx.lower, u_upper=dx.upper, lqr_iter=10, verbose=-1, exit_unconverged=False,
                   detach_unconverged=False, linesearch_decay=dx.linesearch_decay, max_linesearch_iter=dx.max_linesearch_iter,
                   grad_method=mpc.GradMethods.AUTO_DIFF, eps=1e-12, backprop=False, not_improved_lim=10 ** 6)
    cost = QuadCost(Q, pp)
for _ in range(5):
    ctrl(x0, cost, dx)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    ctrl(x0, cost, dx)
torch.cuda.synchronize()
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(28)
print(out.getvalue()[:6000])
