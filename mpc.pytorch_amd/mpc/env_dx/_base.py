"""Common part of the shipped simulators: batching conventions, the quadratic objective helper and
the bridge to the device kernels."""
import torch
from torch import nn

from .. import _native


class SimulatorDx(nn.Module):
    """x [B,n_state] (or [n_state]), u [B,1] (or [1]) -> next state.  Subclasses set n_state,
    n_ctrl, dt, lower/upper, goal_state, goal_weights, ctrl_penalty, mpc_eps, linesearch_decay,
    max_linesearch_iter and implement `_transition(x, u_clamped, params)`."""

    _kind = None        # _native.ENV_*
    _u_max = None

    def forward(self, x, u):
        single = x.dim() == 1
        if single:
            x, u = x.unsqueeze(0), u.unsqueeze(0)
        assert x.dim() == 2 and u.dim() == 2
        assert x.shape[0] == u.shape[0]
        assert x.shape[1] == self.n_state and u.shape[1] == self.n_ctrl
        if self.params.device != x.device:
            self.params = self.params.to(x.device)
        nxt = self._transition(x, u[:, 0].clamp(-self._u_max, self._u_max), self.params)
        return nxt.squeeze(0) if single else nxt

    def get_true_obj(self):
        """(q, p) of the objective 0.5 tau' diag(q) tau + p' tau the examples hand to MPC."""
        q = torch.cat((self.goal_weights, self.ctrl_penalty * torch.ones(self.n_ctrl)))
        assert not hasattr(self, "mpc_lin")
        p = torch.cat((-self.goal_weights.sqrt() * self.goal_state, torch.zeros(self.n_ctrl)))
        return q, p

    # ---- device bridge ---------------------------------------------------------------------
    def native_env(self):
        """EnvSpec for the kernels (include/mpc_lqr.h: mpc_env_dynamics)."""
        return _native.EnvSpec(self._kind, self.params, self.dt, self._u_max)

    def grad_input(self, x, u):
        """R = d f/dx [N,ns,ns], S = d f/du [N,ns,1] at N points, closed form, one kernel
        (what GradMethods.ANALYTIC asks of a dynamics module, mpc/mpc.py:504).  Not differentiable
        w.r.t. the parameters -- use GradMethods.AUTO_DIFF when learning them."""
        F, _ = _native.backend().env_linearize(self.native_env(), x, u)
        ns = self.n_state
        return F[:, :, :ns], F[:, :, ns:]

    def _figure(self, ax, lim):
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        if ax is None:
            fig, ax = plt.subplots(figsize=(6, 6))
        else:
            fig = ax.get_figure()
        ax.set_xlim((-lim, lim))
        ax.set_ylim((-lim, lim))
        return fig, ax
