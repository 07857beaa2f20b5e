"""Dynamics modules with closed-form input Jacobians: `AffineDynamics`, `NNDynamics`,
`CtrlPassthroughDynamics` -- the host-side mirror of the reference's mpc/dynamics.py (same class
names, constructor arguments and `forward` / `grad_input` contracts; :15-130, :133-158, :161-202).

They sit on either side of the LQR step: `forward` is what a module rollout calls once per timestep
(mpc/lqr_step.py:223-225), `grad_input` is what GradMethods.ANALYTIC linearises with
(mpc/mpc.py:495-512).  Plain torch; they run wherever their tensors live.
"""
import torch
from torch import nn

_ACTIVATIONS = {
    "sigmoid": torch.sigmoid,
    "relu": torch.relu,
    "elu": nn.functional.elu,
}
ACTS = _ACTIVATIONS      # the reference exports the table under this name


def _batched(*tensors):
    """Promote 1-D arguments to a batch of one; report whether the first one was 1-D."""
    was_vector = tensors[0].dim() == 1
    return was_vector, tuple(t.unsqueeze(0) if t.dim() == 1 else t for t in tensors)


class AffineDynamics(nn.Module):
    """x' = A x + B u + c with one (A, B, c) shared by the batch.  A [ns,ns], B [ns,nc], c [ns] or None."""

    def __init__(self, A, B, c=None):
        super().__init__()
        assert A.dim() == 2
        assert B.dim() == 2
        assert c is None or c.dim() == 1
        self.A, self.B, self.c = A, B, c

    def forward(self, x, u):
        was_vector, (x, u) = _batched(x, u)
        nxt = nn.functional.linear(x, self.A) + nn.functional.linear(u, self.B)
        if self.c is not None:
            nxt = nxt + self.c
        return nxt[0] if was_vector else nxt

    def grad_input(self, x, u):
        batch = x.shape[0]
        return self.A.expand(batch, *self.A.shape).clone(), self.B.expand(batch, *self.B.shape).clone()


class NNDynamics(nn.Module):
    """A fully connected network [x;u] -> x' (plus x itself when `passthrough`).

    `grad_input(x, u)` returns (d x'/dx, d x'/du) at the points of the most recent `forward` call: it
    re-uses that call's hidden activations instead of re-evaluating the network (the reference does the
    same), and knows the derivative of 'relu' and 'sigmoid' layers.
    """

    def __init__(self, n_state, n_ctrl, hidden_sizes=[100], activation="sigmoid", passthrough=True):
        super().__init__()
        assert activation in _ACTIVATIONS
        self.activation, self.passthrough = activation, passthrough
        sizes = (n_state + n_ctrl, *hidden_sizes, n_state)
        self.fcs = nn.ModuleList([nn.Linear(fan_in, fan_out) for fan_in, fan_out in zip(sizes, sizes[1:])])
        self._wire()

    def _wire(self):
        """(Re)build the views the reference exposes: per-layer activations `acts`, weights `Ws`."""
        hidden = len(self.fcs) - 1
        self.acts = [_ACTIVATIONS[self.activation]] * hidden + [lambda z: z]
        self._stock_acts = list(self.acts)        # native_net: a caller that swaps an activation keeps the module path
        self.Ws = [layer.weight for layer in self.fcs]
        self.zs = []

    # pickles carry only what cannot be rebuilt
    def __getstate__(self):
        return (self.fcs, self.activation, self.passthrough)

    def __setstate__(self, state):
        super().__init__()
        self.fcs, self.activation = state[0], state[1]
        self.passthrough = state[2] if len(state) > 2 else True      # older pickles had no flag
        self._wire()

    def forward(self, x, u):
        was_vector, (x, u) = _batched(x, u)
        z, kept = torch.cat((x, u), dim=1), []
        for act, layer in zip(self.acts, self.fcs):
            z = act(layer(z))
            kept.append(z)
        self.zs = kept[:-1]                       # the output layer is linear: nothing to remember
        out = z + x if self.passthrough else z
        return out[0] if was_vector else out

    def native_net(self, like):
        """The network as the kernels take it (csrc/nn_dynamics.hip: layers on MFMA, 16 problems per wavefront), or None
        when this network / tensor is outside their limits (fp32 on the device, <= 4 layers, n_state <= 16) -- the
        caller then calls the module timestep by timestep like the reference (mpc/lqr_step.py:223-225)."""
        from ._native import MlpSpec
        # the kernels rebuild the computation from fcs / activation / passthrough: a subclass that overrides forward
        # (input normalisation, extra terms), a changed self.acts or a registered forward hook would be bypassed
        if type(self).forward is not NNDynamics.forward or self._forward_hooks or self._forward_pre_hooks:
            return None
        if len(self.acts) != len(self.fcs) or any(a is not b for a, b in zip(self.acts, self._stock_acts)):
            return None
        weights = [layer.weight for layer in self.fcs]
        if not MlpSpec.supported(weights, self.activation, like):
            return None
        self.zs = []          # (the kernels do not fill the activations grad_input re-uses: a stale set must not pass for a fresh one)
        return MlpSpec(weights, [layer.bias for layer in self.fcs], self.activation, self.passthrough)

    def _slope(self, z):
        if self.activation == "relu":
            return (z > 0).to(z.dtype)
        if self.activation == "sigmoid":
            return z * (1. - z)
        assert False, "grad_input knows relu and sigmoid"

    def grad_input(self, x, u):
        was_vector = x.dim() == 1
        n_state = x.shape[-1]
        keep_graph = torch.is_grad_enabled()
        weights = self.Ws if keep_graph else [W.detach() for W in self.Ws]
        hidden = self.zs if keep_graph else [z.detach() for z in self.zs]
        assert len(hidden) == len(weights) - 1
        # forward accumulation from the input side: J <- diag(act'(z_l)) W_l J
        jac = None
        for W, z in zip(weights[:-1], hidden):
            layer_jac = self._slope(z).unsqueeze(2) * W                     # [B, out, in]
            jac = layer_jac if jac is None else layer_jac.bmm(jac)
        last = weights[-1]
        jac = last.expand(hidden[0].shape[0] if hidden else x.reshape(-1, n_state).shape[0], *last.shape) \
            if jac is None else torch.matmul(last, jac)
        R, S = jac[..., :n_state], jac[..., n_state:]
        if self.passthrough:
            R = R + torch.eye(n_state, dtype=R.dtype, device=R.device)
        if was_vector:
            R, S = R[0], S[0]
        return R, S


class CtrlPassthroughDynamics(nn.Module):
    """Dynamics of the slew-rate augmentation: the state is (previous control, x) and a step returns
    (this control, dynamics(x, u))."""

    def __init__(self, dynamics):
        super().__init__()
        self.dynamics = dynamics

    def forward(self, tilde_x, u):
        was_vector, (tilde_x, u) = _batched(tilde_x, u)
        inner_next = self.dynamics(tilde_x[:, u.shape[1]:], u)
        out = torch.cat((u, inner_next), dim=1)
        return out.squeeze() if was_vector else out

    def native_net(self, like):
        """Around an NNDynamics the kernels can run (fp32, augmented n_state <= 16): the augmented network, else None."""
        if type(self).forward is not CtrlPassthroughDynamics.forward or self._forward_hooks or self._forward_pre_hooks:
            return None
        inner = getattr(self.dynamics, "native_net", None)
        net = inner(like) if inner is not None and not isinstance(self.dynamics, CtrlPassthroughDynamics) else None
        if net is None or net.n_state + net.n_ctrl > 16:
            return None
        return net.augmented()

    def grad_input(self, x, u):
        assert False, "Unimplemented"
