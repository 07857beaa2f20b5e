"""Learnable / affine dynamics modules with closed-form input Jacobians -- host-side mirror of the
reference's mpc/dynamics.py (NNDynamics :15-130, CtrlPassthroughDynamics :133-158,
AffineDynamics :161-202).  These are callers of the LQR step (their `forward` is the module rollout
of mpc/lqr_step.py:223-225, their `grad_input` feeds GradMethods.ANALYTIC, mpc/mpc.py:495-512);
they are plain torch modules and run wherever their tensors live."""
import torch
import torch.nn.functional as nnF
from torch import nn

ACTS = {
    "sigmoid": torch.sigmoid,
    "relu": nnF.relu,
    "elu": nnF.elu,
}


def _identity(z):
    return z


class NNDynamics(nn.Module):
    """MLP x_{t+1} = net([x;u]) (+ x if passthrough).  `grad_input` returns the Jacobians
    R = d/dx, S = d/du at the points of the LAST forward call (it re-uses that call's hidden
    activations, as the reference does) for 'relu' and 'sigmoid' activations."""

    def __init__(self, n_state, n_ctrl, hidden_sizes=[100], activation="sigmoid", passthrough=True):
        super().__init__()
        assert activation in ACTS
        self.passthrough = passthrough
        self.activation = activation
        widths = [n_state + n_ctrl] + list(hidden_sizes) + [n_state]
        self.fcs = nn.ModuleList(nn.Linear(a, b) for a, b in zip(widths[:-1], widths[1:]))
        self._wire()

    def _wire(self):
        self.acts = [ACTS[self.activation]] * (len(self.fcs) - 1) + [_identity]
        self.Ws = [fc.weight for fc in self.fcs]
        self.zs = []

    def __getstate__(self):
        return (self.fcs, self.activation, self.passthrough)

    def __setstate__(self, state):
        super().__init__()
        if len(state) == 2:          # pickles written before `passthrough` existed
            self.fcs, self.activation = state
            self.passthrough = True
        else:
            self.fcs, self.activation, self.passthrough = state
        self._wire()

    def forward(self, x, u):
        single = x.dim() == 1
        if single:
            x = x.unsqueeze(0)
        if u.dim() == 1:
            u = u.unsqueeze(0)
        z = torch.cat((x, u), 1)
        hidden = []
        for act, fc in zip(self.acts, self.fcs):
            z = act(fc(z))
            hidden.append(z)
        self.zs = hidden[:-1]         # hidden activations only; the output layer is linear
        if self.passthrough:
            z = z + x
        return z.squeeze(0) if single else z

    def grad_input(self, x, u):
        single = x.dim() == 1
        n_batch, n_state = (1, x.shape[0]) if single else x.shape
        diff = x.requires_grad or u.requires_grad or torch.is_grad_enabled()
        Ws = self.Ws if diff else [W.detach() for W in self.Ws]
        zs = self.zs if diff else [z.detach() for z in self.zs]
        assert len(zs) == len(Ws) - 1
        jac = Ws[-1].unsqueeze(0).expand(n_batch, -1, -1)
        for W, z in zip(reversed(Ws[:-1]), reversed(zs)):
            if self.activation == "relu":
                slope = (z > 0).to(W.dtype)
            elif self.activation == "sigmoid":
                slope = z * (1. - z)
            else:
                assert False
            jac = jac.bmm(slope.unsqueeze(2) * W.unsqueeze(0))
        R, S = jac[:, :, :n_state], jac[:, :, n_state:]
        if self.passthrough:
            R = R + torch.eye(n_state, dtype=R.dtype, device=R.device).unsqueeze(0)
        if single:
            R, S = R.squeeze(0), S.squeeze(0)
        return R, S


class CtrlPassthroughDynamics(nn.Module):
    """Augmented dynamics for slew-rate problems: state (u_prev, x) -> (u, dynamics(x, u))."""

    def __init__(self, dynamics):
        super().__init__()
        self.dynamics = dynamics

    def forward(self, tilde_x, u):
        single = tilde_x.dim() == 1
        if single:
            tilde_x = tilde_x.unsqueeze(0)
        if u.dim() == 1:
            u = u.unsqueeze(0)
        nxt = torch.cat((u, self.dynamics(tilde_x[:, u.shape[1]:], u)), dim=1)
        return nxt.squeeze() if single else nxt

    def grad_input(self, x, u):
        assert False, "Unimplemented"


class AffineDynamics(nn.Module):
    """x_{t+1} = A x + B u (+ c), one (A, B, c) for the whole batch."""

    def __init__(self, A, B, c=None):
        super().__init__()
        assert A.dim() == 2
        assert B.dim() == 2
        if c is not None:
            assert c.dim() == 1
        self.A, self.B, self.c = A, B, c

    def forward(self, x, u):
        single = x.dim() == 1
        if single:
            x = x.unsqueeze(0)
        if u.dim() == 1:
            u = u.unsqueeze(0)
        z = x.mm(self.A.t()) + u.mm(self.B.t())
        if self.c is not None:
            z = z + self.c
        return z.squeeze(0) if single else z

    def grad_input(self, x, u):
        n_batch = x.shape[0]
        return (self.A.unsqueeze(0).repeat(n_batch, 1, 1), self.B.unsqueeze(0).repeat(n_batch, 1, 1))
