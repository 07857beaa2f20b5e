#!/usr/bin/env python3
"""Receding-horizon control of a batch of torque-limited pendulums (the reference's
mpc/env_dx/control.py, without the video): each control step solves a T-step iLQR problem for every
pendulum of the batch on the device, applies the first control and warm-starts the next solve.

    python examples/pendulum_control.py [n_batch]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mpc.pytorch_amd"))
from mpc import mpc                                    # noqa: E402
from mpc.mpc import QuadCost, GradMethods              # noqa: E402
from mpc.env_dx import pendulum                        # noqa: E402

dev = "cuda:0"
n_batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T, steps = 20, 40
dx = pendulum.PendulumDx()
th = torch.linspace(-2.5, 2.5, n_batch)
x = torch.stack((th.cos(), th.sin(), torch.zeros(n_batch)), 1).to(dev)       # hanging or tilted, at rest
q, p = dx.get_true_obj()
Q = torch.diag(q).repeat(T, n_batch, 1, 1).to(dev)
pp = p.repeat(T, n_batch, 1).to(dev)

def controller(u_init):
    return mpc.MPC(dx.n_state, dx.n_ctrl, T, u_lower=dx.lower, u_upper=dx.upper, u_init=u_init,
                   lqr_iter=50 if u_init is None else 5, verbose=-1, exit_unconverged=False,
                   detach_unconverged=False, linesearch_decay=dx.linesearch_decay,
                   max_linesearch_iter=dx.max_linesearch_iter, grad_method=GradMethods.AUTO_DIFF, eps=1e-4)


_, us0, _ = controller(None)(x, QuadCost(Q, pp), dx)        # first calls: load the kernels (ours and torch's)
dx(x, us0[0])
u_init = None
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(steps):
    xs, us, costs = controller(u_init)(x, QuadCost(Q, pp), dx)
    x = dx(x, us[0]).detach()                                                  # apply the first control
    u_init = torch.cat((us[1:], us[-1:]), 0).detach()                          # shift the plan
torch.cuda.synchronize()
dt = time.perf_counter() - t0
up = float((x[:, 0] > 0.95).float().mean())
print("%d pendulums x %d control steps in %.2f s (%.1f ms per receding-horizon step); %.0f %% upright at the end"
      % (n_batch, steps, dt, 1e3 * dt / steps, 100 * up))
