"""Host-side mirror of the reference's `mpc/util.py` helpers that sit on the LQR hot path.

`get_traj` / `get_cost` for LinDx / QuadCost run as one fused kernel (mpc_traj_cost); the small
batched-algebra helpers (bmv, bger, ...) exist only so user code written against the reference
keeps importing -- the kernels never call them.
"""
import operator

import torch

from . import _native


def bmv(X, y):
    """batched matrix-vector product [B,m,n] x [B,n] -> [B,m]   (reference mpc/util.py:44-45)"""
    return (X * y.unsqueeze(1)).sum(2)      # elementwise: a batched GEMM of tiny matrices is far slower


def bger(x, y):
    """batched outer product [B,m],[B,n] -> [B,m,n]   (reference mpc/util.py:40-41)"""
    return torch.einsum("bi,bj->bij", x, y)


def bquad(x, Q):
    """batched quadratic form x'Qx -> [B]   (reference mpc/util.py:48-49)"""
    return torch.einsum("bi,bij,bj->b", x, Q, x)


def bdot(x, y):
    """batched dot product -> [B]   (reference mpc/util.py:52-53)"""
    return (x * y).sum(1)


def bdiag(d):
    """[B,n] -> [B,n,n] diagonal matrices   (reference mpc/util.py:30-37)"""
    return torch.diag_embed(d)


def eclamp(x, lower, upper):
    """Element-wise clamp with float or same-shape tensor bounds, IN PLACE like the reference
    (mpc/util.py:56-70)."""
    lo = lower if torch.is_tensor(lower) else torch.full_like(x, float(lower))
    hi = upper if torch.is_tensor(upper) else torch.full_like(x, float(upper))
    assert lo.shape == x.shape and hi.shape == x.shape
    x.copy_(torch.min(torch.max(x, lo), hi))
    return x


def detach_maybe(x):
    if x is None:
        return None
    return x.detach() if x.requires_grad else x


def get_data_maybe(x):
    return x.detach() if torch.is_tensor(x) else x


data_maybe = detach_maybe


def data_maybe(x):
    """x.data, or None (reference mpc/util.py:162-165)"""
    return None if x is None else x.detach()


def jacobian(f, x, eps):
    """Central-difference Jacobian of f at the single point x ([n] or [1,n]) -> [m, n]
    (reference mpc/util.py:8-18).  All 2n evaluations go through f in one batch when f accepts a
    leading batch axis, else one by one like the reference."""
    if x.ndimension() == 2:
        assert x.size(0) == 1
        x = x.squeeze(0)
    n = len(x)
    e = eps * torch.eye(n, dtype=x.dtype, device=x.device)
    cols = [(f(x + e[i]) - f(x - e[i])) / (2. * eps) for i in range(n)]
    return torch.stack(cols).transpose(0, 1)


def expandParam(X, n_batch, nDim):
    if X.ndimension() in (0, nDim):
        return X, False
    if X.ndimension() == nDim - 1:
        return X.unsqueeze(0).expand(n_batch, *X.shape), True
    raise RuntimeError("Unexpected number of dimensions.")


_seen_tables = set()


def table_log(tag, cols):
    """Print one row of a `| a | b |` table, header on first use (reference mpc/util.py:77-99)."""
    if tag not in _seen_tables:
        _seen_tables.add(tag)
        print("| " + " | ".join(map(operator.itemgetter(0), cols)) + " |")
    cells = [(c[2].format(c[1]) if len(c) == 3 else str(c[1])) for c in cols]
    print("| " + " | ".join(cells) + " |")


def _is_lin(dynamics):
    from .mpc import LinDx
    return isinstance(dynamics, LinDx)


def get_traj(T, u, x_init, dynamics):
    """Nominal rollout x_{t+1} = F_t [x_t;u_t] + f_t (LinDx) or x_{t+1} = dynamics(x_t,u_t).
    Reference: mpc/util.py:102-126.  Returns [T, B, n_state] (detached)."""
    u = get_data_maybe(u)
    x_init = get_data_maybe(x_init)
    if _is_lin(dynamics):
        F = get_data_maybe(dynamics.F)
        f = get_data_maybe(dynamics.f)
        if f is not None and f.numel() > 0:
            assert f.shape == F.shape[:3]
        x, _ = _native.backend().traj_cost(x_init, u, F, f)
        return x
    if hasattr(dynamics, "native_env") and T > 1:       # a shipped simulator: one kernel
        x, _ = _native.backend().env_traj_cost(x_init, u, dynamics.native_env())
        return x
    net = dynamics.native_net(x_init) if hasattr(dynamics, "native_net") and T > 1 else None
    if net is not None:                                  # NNDynamics: the whole rollout in one kernel
        x, _ = _native.backend().mlp_traj_cost(x_init, u.to(x_init.dtype), net)
        return x
    xs = [x_init]
    with torch.no_grad():
        for t in range(T - 1):
            xs.append(dynamics(xs[t], u[t]).detach())
    return torch.stack(xs, 0)


def get_cost(T, u, cost, dynamics=None, x_init=None, x=None):
    """sum_t 0.5 tau'C tau + c'tau (QuadCost) or sum_t cost(tau_t).  Reference: mpc/util.py:129-153."""
    from .mpc import QuadCost
    assert x_init is not None or x is not None
    u = get_data_maybe(u)
    if isinstance(cost, QuadCost) and x is None and _is_lin(dynamics):
        _, tot = _native.backend().traj_cost(get_data_maybe(x_init), u, get_data_maybe(dynamics.F),
                                             get_data_maybe(dynamics.f), get_data_maybe(cost.C),
                                             get_data_maybe(cost.c), want_x=False)
        return tot
    if x is None:
        x = get_traj(T, u, x_init, dynamics)
    tau = torch.cat((get_data_maybe(x), u), 2)
    if isinstance(cost, QuadCost):
        C, c = get_data_maybe(cost.C), get_data_maybe(cost.c)
        return (0.5 * torch.einsum("tbi,tbij,tbj->tb", tau, C, tau) + (tau * c).sum(2)).sum(0)
    with torch.no_grad():
        return torch.stack([cost(tau[t]) for t in range(T)], 0).sum(0)
