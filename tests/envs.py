"""TEST FIXTURES: the two simulator dynamics BASELINE.json's configs 2 and 3 run iLQR on, written
fresh as plain torch modules (the product does not ship environments -- SURVEY.md section 2, rows
10-11; the reference's own are mpc/env_dx/pendulum.py:49-84 and cartpole.py:63-96).  Differentiable,
no `grad_input`, so MPC linearises them with GradMethods.AUTO_DIFF like the reference's demos do."""
import torch
from torch import nn


class PendulumSim(nn.Module):
    """state (cos th, sin th, dth), control torque in [-2, 2]; g = 10, m = l = 1, dt = 0.05."""
    n_state, n_ctrl = 3, 1
    lower, upper = -2.0, 2.0

    def forward(self, x, u):
        g, m, l, dt = 10.0, 1.0, 1.0, 0.05
        tq = u[:, 0].clamp(-2.0, 2.0)
        c, s, w = x[:, 0], x[:, 1], x[:, 2]
        th = torch.atan2(s, c)
        w2 = w + dt * (1.5 * g / l * s + 3.0 * tq / (m * l * l))
        th2 = th + dt * w2
        return torch.stack((torch.cos(th2), torch.sin(th2), w2), 1)

    @staticmethod
    def objective(dtype):
        """(q, p) of the quadratic objective 0.5 tau' diag(q) tau + p' tau (goal: upright, at rest)"""
        w = torch.tensor([1.0, 1.0, 0.1], dtype=dtype)
        goal = torch.tensor([1.0, 0.0, 0.0], dtype=dtype)
        return torch.cat((w, torch.full((1,), 0.001, dtype=dtype))), torch.cat((-w.sqrt() * goal, torch.zeros(1, dtype=dtype)))


class CartpoleSim(nn.Module):
    """state (x, dx, cos th, sin th, dth), force in [-100, 100]; g = 9.8, cart 1.0, pole 0.1, l = 0.5, dt = 0.05."""
    n_state, n_ctrl = 5, 1
    lower, upper = -100.0, 100.0

    def forward(self, st, u):
        g, mc, mp, l, dt = 9.8, 1.0, 0.1, 0.5, 0.05
        mt, pml = mc + mp, mp * l
        f = u[:, 0].clamp(-100.0, 100.0)
        x, v, c, s, w = st.unbind(1)
        th = torch.atan2(s, c)
        cart_in = (f + pml * w * w * s) / mt
        th_acc = (g * s - c * cart_in) / (l * (4.0 / 3.0 - mp * c * c / mt))
        x_acc = cart_in - pml * th_acc * c / mt
        x2, v2, th2, w2 = x + dt * v, v + dt * x_acc, th + dt * w, w + dt * th_acc
        return torch.stack((x2, v2, torch.cos(th2), torch.sin(th2), w2), 1)

    @staticmethod
    def objective(dtype):
        w = torch.tensor([0.1, 0.1, 1.0, 1.0, 0.1], dtype=dtype)
        goal = torch.tensor([0.0, 0.0, 1.0, 0.0, 0.0], dtype=dtype)
        return torch.cat((w, torch.full((1,), 0.001, dtype=dtype))), torch.cat((-w.sqrt() * goal, torch.zeros(1, dtype=dtype)))


class SmoothCost(nn.Module):
    """A smooth, convex, NON-quadratic stage cost on tau = (x, u) [B, n] -> [B]:
        0.5 sum_i w_i (tau_i - goal_i)^2 + beta sum_j log cosh((P tau)_j)
    Hessian diag(w) + beta P' diag(sech^2(P tau)) P is positive definite and changes along the trajectory, so
    MPC.approximate_cost (reference mpc/mpc.py:447-487) has something to expand at every iteration.  `goal`
    and `P` are Parameters: the backward of a solve reaches them through the differentiable expansion."""

    def __init__(self, n, m=3, beta=2.0, seed=0, dtype=torch.float64):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.w = (0.5 + torch.rand(n, generator=g, dtype=dtype))
        self.goal = nn.Parameter(0.5 * torch.randn(n, generator=g, dtype=dtype))
        self.P = nn.Parameter(torch.randn(m, n, generator=g, dtype=dtype) / n ** 0.5)
        self.beta = beta

    def forward(self, tau):
        w = self.w.to(tau.device)
        z = tau.matmul(self.P.t())
        # log cosh z, written so that large |z| does not overflow
        lc = z.abs() + torch.log1p(torch.exp(-2.0 * z.abs())) - 0.6931471805599453
        return 0.5 * (w * (tau - self.goal) ** 2).sum(1) + self.beta * lc.sum(1)
