#!/usr/bin/env python3
"""BASELINE.json configs 2 and 3: whole MPC.forward (iLQR) on the shipped simulators, fp32.
kernel path (mpc.env_dx modules: linearisation kernel + simulator inside the rollout kernel) beside
the host-driven module path (plain torch module: autograd linearisation, per-timestep rollout)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mpc import mpc                                   # noqa: E402
from mpc.mpc import QuadCost                          # noqa: E402
from mpc.env_dx import cartpole, pendulum             # noqa: E402
import envs                                           # noqa: E402

DEV = "cuda:0"


def problem(kind, B, T):
    g = torch.Generator().manual_seed(0)
    if kind == "pendulum":
        dx, plain = pendulum.PendulumDx(), envs.PendulumSim()
        th = (torch.rand(B, generator=g) - 0.5) * np.pi
        x0 = torch.stack((th.cos(), th.sin(), (torch.rand(B, generator=g) - 0.5) * 2), 1)
    else:
        dx, plain = cartpole.CartpoleDx(), envs.CartpoleSim()
        th = (torch.rand(B, generator=g) - 0.5) * 0.6
        zz = 0.2 * torch.randn(B, 3, generator=g)
        x0 = torch.stack((zz[:, 0], zz[:, 1], th.cos(), th.sin(), zz[:, 2]), 1)
    q, p = dx.get_true_obj()
    return dx, plain, x0.to(DEV), torch.diag(q).repeat(T, B, 1, 1).to(DEV), p.repeat(T, B, 1).to(DEV)


def run(kind, B, T, iters, reps):
    dx, plain, x0, Q, pp = problem(kind, B, T)
    out = {"config": kind, "B": B, "T": T, "lqr_iter": iters}
    legs = (("kernel_path", dx, reps), ("module_path", plain, max(1, reps // 5)))
    if "--kernel-only" in sys.argv:
        legs = legs[:1]
    for label, mod, r in legs:
        ctrl = mpc.MPC(dx.n_state, 1, T, u_lower=dx.lower, u_upper=dx.upper, lqr_iter=iters, verbose=-1,
                       exit_unconverged=False, detach_unconverged=False, linesearch_decay=dx.linesearch_decay,
                       max_linesearch_iter=dx.max_linesearch_iter, grad_method=mpc.GradMethods.AUTO_DIFF,
                       eps=1e-12, backprop=False, not_improved_lim=10 ** 6)
        ctrl(x0, QuadCost(Q, pp), mod)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(r):
            x, u, c = ctrl(x0, QuadCost(Q, pp), mod)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / r * 1e3
        out[label] = {"ms_per_solve": round(ms, 3), "ms_per_ilqr_iteration": round(ms / iters, 4),
                      "problem_steps_per_s": round(B * T * iters / (ms * 1e-3)), "mean_cost": float(c.mean())}
    if "module_path" in out:
        out["speedup"] = round(out["module_path"]["ms_per_solve"] / out["kernel_path"]["ms_per_solve"], 1)
    return out


if __name__ == "__main__":
    res = [run("pendulum", 1024, 20, 10, 10), run("cartpole", 4096, 25, 10, 10)]
    for r in res:
        print(json.dumps(r))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "bench_ilqr_env.json"), "w"), indent=1)
