#!/usr/bin/env python3
"""Identify a dynamics network through the controller (the setting mpc.pytorch was written for): an `NNDynamics`
is trained so that MPC on the NETWORK reproduces the controls an expert computes on the true pendulum -- the loss
is on the controller's output, the gradient reaches the network's weights through the KKT backward of the LQR step.

Every MPC.forward here runs the network inside the kernels (trajectory, analytic linearisation, line-searched
rollout: csrc/nn_dynamics.hip); only the last, differentiable linearisation of a solve goes through autograd.

    python examples/learned_dynamics.py [n_batch] [epochs]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mpc.pytorch_amd"))
from mpc import mpc                                    # noqa: E402
from mpc.mpc import QuadCost, GradMethods              # noqa: E402
from mpc.dynamics import NNDynamics                    # noqa: E402
from mpc.env_dx import pendulum                        # noqa: E402

dev = "cuda:0"
n_batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
T = 10
torch.manual_seed(0)
true_dx = pendulum.PendulumDx()
q, p = true_dx.get_true_obj()
Q = torch.diag(q).repeat(T, n_batch, 1, 1).to(dev)
pp = p.repeat(T, n_batch, 1).to(dev)
cost = QuadCost(Q, pp)


def controller(grad_method, lqr_iter):
    return mpc.MPC(true_dx.n_state, true_dx.n_ctrl, T, u_lower=true_dx.lower, u_upper=true_dx.upper, lqr_iter=lqr_iter,
                   verbose=-1, exit_unconverged=False, detach_unconverged=False, grad_method=grad_method,
                   linesearch_decay=true_dx.linesearch_decay, max_linesearch_iter=true_dx.max_linesearch_iter)


def sample_states(n):
    th = (torch.rand(n) - 0.5) * 2.0
    return torch.stack((th.cos(), th.sin(), (torch.rand(n) - 0.5)), 1).to(dev)


net = NNDynamics(true_dx.n_state, true_dx.n_ctrl, hidden_sizes=[64], activation="sigmoid").to(dev)
opt = torch.optim.Adam(net.parameters(), lr=3e-3)
t0 = time.time()
for epoch in range(epochs):
    x0 = sample_states(n_batch)
    with torch.no_grad():
        _, u_expert, _ = controller(GradMethods.AUTO_DIFF, 15)(x0, cost, true_dx)       # the simulator inside the kernels
    _, u_net, _ = controller(GradMethods.ANALYTIC, 15)(x0, cost, net)                   # the network inside the kernels
    # one-step model error keeps the network honest where the controller does not look
    with torch.no_grad():
        x_next = true_dx(x0, u_expert[0])
    loss = (u_net - u_expert).pow(2).mean() + (net(x0, u_expert[0]) - x_next).pow(2).mean()
    opt.zero_grad()
    loss.backward()
    opt.step()
    if epoch % 5 == 0 or epoch + 1 == epochs:
        print("epoch %3d  imitation + model loss %.5f   (%.1f s)" % (epoch, float(loss.detach()), time.time() - t0))
print("done: %d solves of %d problems each" % (2 * epochs, n_batch))
