#!/usr/bin/env python3
"""Small-batch latency of MPC.forward (control-loop use): pendulum and the headline shape at B = 1, 8."""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import bench
from mpc import mpc
from mpc.mpc import QuadCost, LinDx, GradMethods
from mpc.env_dx import pendulum
out = {}
for B in (1, 8):
    dx = pendulum.PendulumDx(); T = 20
    th = torch.linspace(-1, 1, B); x0 = torch.stack((th.cos(), th.sin(), torch.zeros(B)), 1).cuda()
    q, p = dx.get_true_obj(); Q = torch.diag(q).repeat(T, B, 1, 1).cuda(); pp = p.repeat(T, B, 1).cuda()
    ctrl = mpc.MPC(3, 1, T, u_lower=-2., u_upper=2., lqr_iter=5, verbose=-1, exit_unconverged=False, detach_unconverged=False,
                   linesearch_decay=0.2, max_linesearch_iter=5, grad_method=GradMethods.AUTO_DIFF, eps=1e-12, not_improved_lim=100)
    pr = bench.make_problem(12, 4, 50, B, torch.float32, "cuda:0", seed=1)
    c2 = mpc.MPC(12, 4, 50, lqr_iter=5, verbose=-1, exit_unconverged=False, detach_unconverged=False, eps=1e-12, not_improved_lim=100)
    for name, fn in (("pendulum", lambda: ctrl(x0, QuadCost(Q, pp), dx)), ("ns12_nc4_T50", lambda: c2(pr["x_init"], QuadCost(pr["C"], pr["c"]), LinDx(pr["F"], pr["f"])))):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize()
        out["%s_B%d_5iter_us" % (name, B)] = round(1e6 * (time.perf_counter() - t0) / 20, 1)
print(json.dumps(out))
