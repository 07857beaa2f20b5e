"""PendulumDx -- the torque-limited pendulum of the reference (mpc/env_dx/pendulum.py:18-115).

State (cos th, sin th, dth); control = torque, clamped to +-max_torque inside the dynamics.
`simple=True`: parameters (g, m, l); `simple=False` adds damping d and a gravity bias b."""
import torch

from .. import _native
from ._base import SimulatorDx


class PendulumDx(SimulatorDx):
    def __init__(self, params=None, simple=True):
        super().__init__()
        self.simple = simple
        self.max_torque = 2.0
        self.dt = 0.05
        self.n_state, self.n_ctrl = 3, 1
        if params is None:
            params = torch.tensor((10., 1., 1.) if simple else (10., 1., 1., 0., 0.))
        self.params = params
        assert len(self.params) == (3 if simple else 5)
        self.goal_state = torch.tensor([1., 0., 0.])
        self.goal_weights = torch.tensor([1., 1., 0.1])
        self.ctrl_penalty = 0.001
        self.lower, self.upper = -2., 2.
        self.mpc_eps = 1e-3
        self.linesearch_decay = 0.2
        self.max_linesearch_iter = 5

    @property
    def _kind(self):
        return _native.ENV_PENDULUM if self.simple else _native.ENV_PENDULUM_FULL

    @property
    def _u_max(self):
        return self.max_torque

    def _transition(self, x, torque, params):
        c, s, w = x.unbind(1)
        th = torch.atan2(s, c)
        if self.simple:
            g, m, l = params.unbind()
            acc = 1.5 * g / l * s + 3. * torque / (m * l ** 2)
        else:
            g, m, l, d, b = params.unbind()
            acc = 1.5 * g / l * torch.sin(th + b) + 3. * torque / (m * l ** 2) - d * th
        w2 = w + self.dt * acc
        th2 = th + self.dt * w2
        return torch.stack((th2.cos(), th2.sin(), w2), 1)

    def get_frame(self, x, ax=None):
        x = x.detach().reshape(-1).cpu()
        assert len(x) == 3
        l = float(self.params[2])
        fig, ax = self._figure(ax, 1.2 * l)
        ax.plot((0, float(x[1]) * l), (0, float(x[0]) * l), color="k")
        return fig, ax
