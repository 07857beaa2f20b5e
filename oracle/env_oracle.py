"""TEST INFRASTRUCTURE (not product): numpy restatement of the simulator dynamics the reference
ships and of the rollout that calls them.

  pendulum_step   mpc/env_dx/pendulum.py:49-84   (simple and 5-parameter variants)
  cartpole_step   mpc/env_dx/cartpole.py:63-96
  linearize       mpc/mpc.py:490-549: F = d step / d [x;u], f = step(x,u) - F [x;u].  The
                  Jacobian is taken by Richardson-extrapolated central differences in float64 (error ~1e-12), on
                  purpose NOT by the closed form the kernels use, so the two are independent.
  rollout         mpc/lqr_step.py:164-261 with a module as true_dynamics (:223-225) and a QuadCost
                  as true cost (:230-232), one problem at a time (= the reference with n_batch 1).
  mlp_step / mlp_jacobian   mpc/dynamics.py:57-80 (NNDynamics.forward) and :82-128 (grad_input): kind MLP, whose
                  `params` is an `Mlp` (weights, biases, activation, passthrough) instead of a parameter vector.

Parity status: pinned -- tests/test_oracle_golden.py checks all three against outputs of the
reference's own modules (tests/golden/env_*.npz).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline may import this file."""
import numpy as np

PENDULUM, PENDULUM_FULL, CARTPOLE, MLP = 1, 2, 3, 4
DT = 0.05


class Mlp:
    """The network of mpc/dynamics.py:15-36: Linear layers (weights [out, in], biases [out]), the same activation
    after every layer but the last, `passthrough` adds the state to the output (:74-75)."""

    def __init__(self, Ws, bs, activation="sigmoid", passthrough=True):
        self.Ws = [np.asarray(W, dtype=np.float64) for W in Ws]
        self.bs = [np.asarray(b, dtype=np.float64) for b in bs]
        assert activation in ("sigmoid", "relu", "elu")
        self.activation, self.passthrough = activation, bool(passthrough)

    @classmethod
    def from_npz(cls, z):
        L = int(z["nn_meta"][0])
        return cls([z["W%d" % i] for i in range(L)], [z["b%d" % i] for i in range(L)],
                   ("sigmoid", "relu", "elu")[int(z["nn_meta"][1])], bool(z["nn_meta"][2]))


def _act(z, kind):
    if kind == "sigmoid":
        return 1.0 / (1.0 + np.exp(-z))
    if kind == "relu":
        return np.maximum(z, 0.0)
    return np.where(z > 0, z, np.expm1(np.minimum(z, 0.0)))                              # F.elu, alpha = 1


def mlp_hidden(x, u, net):
    """Activations of every hidden layer and the output, at points x [N,ns], u [N,nc] (mpc/dynamics.py:66-72)."""
    z = np.concatenate((np.asarray(x, dtype=np.float64), np.asarray(u, dtype=np.float64)), 1)
    zs = []
    for i, (W, b) in enumerate(zip(net.Ws, net.bs)):
        z = z @ W.T + b
        if i + 1 < len(net.Ws):
            z = _act(z, net.activation)
            zs.append(z)
    return zs, z


def mlp_step(x, u, net):
    _, out = mlp_hidden(x, u, net)
    return out + np.asarray(x, dtype=np.float64) if net.passthrough else out              # :74-75


def mlp_jacobian(x, u, net):
    """d mlp_step / d [x;u] at every point, [N, ns, ns+nc] (grad_input, mpc/dynamics.py:82-128: the chain
    W_L diag(act'(z_{L-1})) W_{L-1} ... with act' = z (1 - z) for the sigmoid and [z > 0] for relu; elu is not
    implemented there (`assert False`), its slope here is 1 / z + 1 on the two branches)."""
    zs, _ = mlp_hidden(x, u, net)
    N, ns = np.asarray(x).shape
    J = None
    for W, z in zip(net.Ws[:-1], zs):
        if net.activation == "sigmoid":
            d = z * (1.0 - z)
        elif net.activation == "relu":
            d = (z > 0).astype(np.float64)
        else:
            d = np.where(z > 0, 1.0, z + 1.0)
        G = d[:, :, None] * W[None]
        J = G if J is None else np.einsum("nij,njk->nik", G, J)
    last = net.Ws[-1]
    J = np.broadcast_to(last, (N,) + last.shape).copy() if J is None else np.einsum("ij,njk->nik", last, J)
    if net.passthrough:
        J[:, :, :ns] += np.eye(ns)
    return J


def u_max_of(kind):
    return 100.0 if kind == CARTPOLE else 2.0


def pendulum_step(x, u, params, simple=True, dt=DT, max_torque=2.0, th=None):
    """x [N,3] = (cos th, sin th, dth), u [N,1].  th: the angle as an argument of its own (linearize differentiates through it by the
    chain rule, not across atan2's branch cut); None: atan2 of the state, as the module has it."""
    x = np.asarray(x, dtype=np.float64)
    tq = np.clip(np.asarray(u, dtype=np.float64)[:, 0], -max_torque, max_torque)      # :66
    c, s, w = x[:, 0], x[:, 1], x[:, 2]
    if th is None:
        th = np.arctan2(s, c)                                                         # :68
    if simple:
        g, m, l = params[:3]
        w2 = w + dt * (-3. * g / (2. * l) * (-s) + 3. * tq / (m * l ** 2))            # :70-71
    else:
        g, m, l, d, b = params
        w2 = w + dt * (-3. * g / (2. * l) * (-np.sin(th + b)) + 3. * tq / (m * l ** 2) - d * th)   # :73-75
    th2 = th + w2 * dt                                                                # :76
    return np.stack((np.cos(th2), np.sin(th2), w2), 1)                                # :77


def cartpole_step(st, u, params, dt=DT, force_mag=100.0, th=None):
    """st [N,5] = (x, dx, cos th, sin th, dth), u [N,1].  th: see pendulum_step."""
    st = np.asarray(st, dtype=np.float64)
    g, mc, mp, l = params
    mt, pml = mp + mc, mp * l                                                         # :70-71
    f = np.clip(np.asarray(u, dtype=np.float64)[:, 0], -force_mag, force_mag)         # :73
    x, v, c, s, w = (st[:, i] for i in range(5))
    if th is None:
        th = np.arctan2(s, c)                                                         # :76
    cart_in = (f + pml * w ** 2 * s) / mt                                             # :78
    th_acc = (g * s - c * cart_in) / (l * (4. / 3. - mp * c ** 2 / mt))               # :79-81
    xacc = cart_in - pml * th_acc * c / mt                                            # :82
    th2 = th + dt * w                                                                 # :86
    return np.stack((x + dt * v, v + dt * xacc, np.cos(th2), np.sin(th2), w + dt * th_acc), 1)   # :84-91


def step(kind, x, u, params, clamp=True, th=None):
    if kind == MLP:
        return mlp_step(x, u, params)
    lim = u_max_of(kind) if clamp else np.inf
    if kind == CARTPOLE:
        return cartpole_step(x, u, params, force_mag=lim, th=th)
    return pendulum_step(x, u, params, simple=(kind == PENDULUM), max_torque=lim, th=th)


def linearize(kind, x, u, params, h=2e-4):
    """x [N,ns], u [N,1] -> F [N,ns,ns+1], f [N,ns]."""
    x = np.asarray(x, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    N, ns = x.shape
    tau = np.concatenate((x, u), 1)
    if kind == MLP:                 # closed form, like the reference's own ANALYTIC path (mpc/mpc.py:495-512)
        F = mlp_jacobian(x, u, params)
        return F, mlp_step(x, u, params) - np.einsum("nij,nj->ni", F, tau)
    F = np.empty((N, ns, ns + 1))

    # The control enters through clamp(u, -u_max, u_max) only; differences are taken of the smooth
    # part at the clamped control and the clamp's own derivative is applied in closed form with
    # autograd's convention (1 on the CLOSED interval) -- a finite difference would return 1/2 on
    # a control sitting exactly at its bound, which is where bounded solves put it.
    lim = u_max_of(kind)
    tau_c = np.concatenate((x, np.clip(u, -lim, lim)), 1)

    # The angle th = atan2(sin, cos) is differentiated by the chain rule (d th = (c ds - s dc) / (c^2 + s^2), autograd's rule), the
    # differences are taken with th as an argument of its own: a difference ACROSS atan2's branch cut (a state within h of
    # th = +-pi) is 2 pi / h times the damping coefficient of the full pendulum model, not a derivative
    # (tools/ref_diff_misc.py found one such point in 30,000).
    ic, isn = (0, 1) if kind != CARTPOLE else (2, 3)
    th0 = np.arctan2(x[:, isn], x[:, ic])

    def central(j, hh):
        e = np.zeros(ns + 1)
        e[j] = hh
        hi = step(kind, (tau_c + e)[:, :ns], (tau_c + e)[:, ns:], params, clamp=False, th=th0)
        lo = step(kind, (tau_c - e)[:, :ns], (tau_c - e)[:, ns:], params, clamp=False, th=th0)
        return (hi - lo) / (2 * hh)

    def central_th(hh):
        hi = step(kind, tau_c[:, :ns], tau_c[:, ns:], params, clamp=False, th=th0 + hh)
        lo = step(kind, tau_c[:, :ns], tau_c[:, ns:], params, clamp=False, th=th0 - hh)
        return (hi - lo) / (2 * hh)
    for j in range(ns + 1):
        F[:, :, j] = (4.0 * central(j, h / 2) - central(j, h)) / 3.0     # Richardson: O(h^4)
    dth = (4.0 * central_th(h / 2) - central_th(h)) / 3.0
    r2 = x[:, ic] ** 2 + x[:, isn] ** 2
    F[:, :, ic] += dth * (-x[:, isn] / r2)[:, None]
    F[:, :, isn] += dth * (x[:, ic] / r2)[:, None]
    F[:, :, ns] *= ((u[:, 0] >= -lim) & (u[:, 0] <= lim))[:, None]
    f = step(kind, x, u, params) - np.einsum("nij,nj->ni", F, tau)
    return F, f


def traj(kind, x_init, u, params):
    """util.get_traj through the simulator (mpc/util.py:107-113)."""
    T = u.shape[0]
    xs = [np.asarray(x_init, dtype=np.float64)]
    for t in range(T - 1):
        xs.append(step(kind, xs[t], u[t], params))
    return np.stack(xs)


def quad_cost(C, c, x, u):
    tau = np.concatenate((x, u), 2)
    return (0.5 * np.einsum("tbi,tbij,tbj->b", tau, C, tau) + np.einsum("tbi,tbi->b", c, tau))


def rollout(kind, params, x_init, C, c, K, k, cur_x, cur_u, lower, upper, decay, max_ls):
    """lqr_forward (mpc/lqr_step.py:164-261), per problem.  Returns new_x, new_u, costs,
    full_du_norm, alphas."""
    T, B, nc = cur_u.shape
    ns = x_init.shape[1]
    old = quad_cost(C, c, cur_x, cur_u)                                               # :169
    new_x = np.empty((T, B, ns)); new_u = np.empty((T, B, nc))
    costs = np.empty(B); full = np.empty(B); alphas = np.empty(B)
    for b in range(B):
        alpha = 1.0
        for it in range(max_ls):
            xs = [x_init[b:b + 1]]
            us = []
            dx = np.zeros((1, ns))
            for t in range(T):
                nu = K[t, b:b + 1] @ dx[0] + cur_u[t, b:b + 1] + alpha * k[t, b:b + 1]    # :192
                if lower is not None:
                    nu = np.clip(nu, lower, upper)                                        # :200-213
                us.append(nu)
                if t < T - 1:
                    nx = step(kind, xs[t], nu, params)                                    # :223-225
                    xs.append(nx)
                    dx = nx - cur_x[t + 1, b:b + 1]                                       # :227
            X, U = np.stack(xs), np.stack(us)
            cost = quad_cost(C[:, b:b + 1], c[:, b:b + 1], X, U)[0]                       # :230-232
            dun = np.sqrt(((cur_u[:, b:b + 1] - U) ** 2).sum())
            if it == 0:
                full[b] = dun                                                             # :243-245
            if cost > old[b] and it + 1 < max_ls:                                         # :176-179, 247
                alpha *= decay
            else:
                break
        new_x[:, b], new_u[:, b], costs[b], alphas[b] = X[:, 0], U[:, 0], cost, alpha
    return new_x, new_u, costs, full, alphas


def _bounds_at(lower, upper, t, ut, delta_u):
    """Bounds of timestep t (floats or [T,B,nc] arrays), intersected with u +- delta_u (mpc/lqr_step.py:200-211)."""
    lo = lower[t] if isinstance(lower, np.ndarray) and lower.ndim == 3 else lower
    hi = upper[t] if isinstance(upper, np.ndarray) and upper.ndim == 3 else upper
    if delta_u is not None:
        lo = np.maximum(lo, ut - delta_u)
        hi = np.minimum(hi, ut + delta_u)
    return lo, hi


def rollout_batched(kind, params, x_init, C, c, K, k, cur_x, cur_u, lower, upper, decay, max_ls, delta_u=None,
                    u_zero_I=None):
    """`rollout` for large batches: the max_ls trials alpha = decay^i are each rolled out over the WHOLE batch
    with numpy and every problem takes its first trial that did not get worse (else the last) -- the per-problem
    loop of `rollout` in a different order, same results (checked against it in tests/test_oracle_golden.py).
    Also returns the per-trial costs and the nominal cost so a test can recognise a line-search tie."""
    T, B, nc = cur_u.shape
    ns = x_init.shape[1]
    old = quad_cost(C, c, cur_x, cur_u)
    done = np.zeros(B, bool)
    new_x = np.empty((T, B, ns)); new_u = np.empty((T, B, nc))
    costs = np.empty(B); full = np.empty(B); alphas = np.empty(B)
    trial_costs = np.full((max_ls, B), np.nan)
    for it in range(max_ls):
        alpha = decay ** it
        xs = [np.asarray(x_init, dtype=np.float64)]
        us = []
        dx = np.zeros((B, ns))
        for t in range(T):
            nu = np.einsum("bij,bj->bi", K[t], dx) + cur_u[t] + alpha * k[t]
            if u_zero_I is not None:
                nu = np.where(u_zero_I[t], 0.0, nu)                                       # :197-198
            if lower is not None:
                lo, hi = _bounds_at(lower, upper, t, cur_u[t], delta_u)
                nu = np.minimum(np.maximum(nu, lo), hi)                                   # util.eclamp
            us.append(nu)
            if t < T - 1:
                nx = step(kind, xs[t], nu, params)
                xs.append(nx)
                dx = nx - cur_x[t + 1]
        X, U = np.stack(xs), np.stack(us)
        cost = quad_cost(C, c, X, U)
        trial_costs[it] = cost
        if it == 0:
            full[:] = np.sqrt(((cur_u - U) ** 2).sum((0, 2)))
        take = ~done & (~(cost > old) | (it + 1 >= max_ls))
        new_x[:, take], new_u[:, take], costs[take], alphas[take] = X[:, take], U[:, take], cost[take], alpha
        done |= take
        if done.all():
            break
    return new_x, new_u, costs, full, alphas, trial_costs, old
