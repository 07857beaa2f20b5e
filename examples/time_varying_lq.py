#!/usr/bin/env python3
"""Time-varying linear-quadratic control with box constraints -- the setting of the reference's
"examples/Time Varying Linear-Quadratic Control" notebook, on an MI355X.

    python examples/time_varying_lq.py

Everything below the imports is what a user of locuslab/mpc.pytorch already has; only the import
path changed."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mpc.pytorch_amd"))
from mpc import mpc                                    # noqa: E402
from mpc.mpc import QuadCost, LinDx                    # noqa: E402

torch.manual_seed(0)
dev = "cuda:0"
n_batch, n_state, n_ctrl, T = 2, 3, 4, 5
n_sc = n_state + n_ctrl

# a random PSD stage cost and time-varying linear dynamics
C = torch.randn(T * n_batch, n_sc, n_sc)
C = torch.bmm(C, C.transpose(1, 2)).view(T, n_batch, n_sc, n_sc)
c = torch.randn(T, n_batch, n_sc)
R = (torch.eye(n_state) + 0.2 * torch.randn(n_state, n_state)).repeat(T, n_batch, 1, 1)
S = torch.randn(T, n_batch, n_state, n_ctrl)
F = torch.cat((R, S), dim=3)
x_init = torch.randn(n_batch, n_state)
u_lower = -torch.rand(T, n_batch, n_ctrl)
u_upper = torch.rand(T, n_batch, n_ctrl)
C, c, F, x_init, u_lower, u_upper = (t.to(dev) for t in (C, c, F, x_init, u_lower, u_upper))
c.requires_grad_(True)

x, u, objs = mpc.MPC(n_state=n_state, n_ctrl=n_ctrl, T=T, u_lower=u_lower, u_upper=u_upper, lqr_iter=20,
                     verbose=1, backprop=True, exit_unconverged=False)(x_init, QuadCost(C, c), LinDx(F))
print("optimal costs per problem:", objs.tolist())
print("first controls:", u[0].tolist())
# the solve is differentiable: d(sum of controls) / d(linear cost term)
u.sum().backward()
print("|d sum(u) / dc| =", float(c.grad.abs().sum()))
