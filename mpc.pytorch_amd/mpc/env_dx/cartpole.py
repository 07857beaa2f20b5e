"""CartpoleDx -- the cart-pole of the reference (mpc/env_dx/cartpole.py:28-127).

State (x, dx, cos th, sin th, dth); control = horizontal force, clamped to +-force_mag inside the
dynamics.  Parameters (gravity, masscart, masspole, length)."""
import math

import torch

from .. import _native
from ._base import SimulatorDx


class CartpoleDx(SimulatorDx):
    _kind = _native.ENV_CARTPOLE

    def __init__(self, params=None):
        super().__init__()
        self.n_state, self.n_ctrl = 5, 1
        if params is None:
            params = torch.tensor((9.8, 1.0, 0.1, 0.5))
        self.params = params
        assert len(self.params) == 4
        self.force_mag = 100.
        self.theta_threshold_radians = math.pi
        self.x_threshold = 2.4
        self.max_velocity = 10
        self.dt = 0.05
        self.lower, self.upper = -self.force_mag, self.force_mag
        self.goal_state = torch.tensor([0., 0., 1., 0., 0.])
        self.goal_weights = torch.tensor([0.1, 0.1, 1., 1., 0.1])
        self.ctrl_penalty = 0.001
        self.mpc_eps = 1e-4
        self.linesearch_decay = 0.5
        self.max_linesearch_iter = 2

    @property
    def _u_max(self):
        return self.force_mag

    def _transition(self, st, force, params):
        g, m_cart, m_pole, l = params.unbind()
        m_tot, pml = m_pole + m_cart, m_pole * l
        px, v, c, s, w = st.unbind(1)
        th = torch.atan2(s, c)
        cart_in = (force + pml * w ** 2 * s) / m_tot
        th_acc = (g * s - c * cart_in) / (l * (4. / 3. - m_pole * c ** 2 / m_tot))
        x_acc = cart_in - pml * th_acc * c / m_tot
        th2 = th + self.dt * w
        return torch.stack((px + self.dt * v, v + self.dt * x_acc, th2.cos(), th2.sin(),
                            w + self.dt * th_acc), 1)

    def get_frame(self, state, ax=None):
        z = state.detach().reshape(-1).cpu()
        assert len(z) == 5
        l = float(self.params[3])
        fig, ax = self._figure(ax, 2 * l)
        ax.plot((float(z[0]), float(z[0]) + float(z[3]) * l), (0, float(z[2]) * l), color="k")
        return fig, ax
