"""TEST-ONLY stand-in for mpc._native.HipBackend, backed by the CPU oracle.

Lets the CPU-only suite exercise the HOST logic of the product package (MPC.forward's iteration /
best-iterate / convergence rules, the autograd wiring of LQRStep, argument expansion) without a
GPU.  Installed with mpc._native.set_backend_for_testing(); the product never imports this file
and has no CPU path of its own.  Per-problem semantics (lockstep=False) like the HIP kernels.
"""
import numpy as np
import torch

from oracle import lqr_oracle as O
from oracle import env_oracle as E


def _np(t):
    return None if t is None else t.detach().cpu().numpy()


def _bound(v):
    if v is None or isinstance(v, float):
        return v
    if isinstance(v, int):
        return float(v)
    return _np(v)


class OracleBackend:
    name = "cpu-oracle (tests only)"

    def __init__(self, lockstep=False, qp_cold=False):
        self.lockstep = lockstep
        self.qp_cold = qp_cold          # every convex box QP from pnqp's own cold start, like the fused float32 kernels up to 12/4 (lqr_oracle.h)
        self.calls = []

    def _t(self, a, like):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dtype=like.dtype)

    def lqr_step(self, x_init, C, c, F, f, cur_x, cur_u, opts, want_gains=False, impl=0, rollout_problem=None, out_x=None, out_u=None):
        assert rollout_problem is None, "oracle backend: split true-cost rollout not modelled"
        self.calls.append("lqr_step")
        T = C.shape[0]
        env0 = getattr(opts, "true_dynamics", None)
        if env0 is not None and getattr(env0, "linearize", False):      # the step linearises the simulator itself
            self.calls.append("inline_linearize")
            ns_, nc_ = x_init.shape[1], C.shape[2] - x_init.shape[1]
            Fl, fl = E.linearize(env0.kind, _np(cur_x[:-1].reshape(-1, ns_)), _np(cur_u[:-1].reshape(-1, nc_)),
                                 _np(env0.params).astype(np.float64))
            F = self._t(Fl, C).view(T - 1, C.shape[1], ns_, ns_ + nc_)
            f = self._t(fl, C).view(T - 1, C.shape[1], ns_)
        Fn = _np(F) if T > 1 else np.zeros((0, C.shape[1], x_init.shape[1], C.shape[2]), _np(C).dtype)
        o = O.lqr_step(_np(x_init), _np(C), _np(c), Fn, _np(f), _np(cur_x), _np(cur_u),
                       _bound(opts.u_lower), _bound(opts.u_upper), _np(opts.u_zero_I), opts.delta_u,
                       opts.linesearch_decay, opts.max_linesearch_iter, lockstep=self.lockstep,
                       return_gains=True, qp_cold=self.qp_cold and not self.lockstep)
        B = C.shape[1]
        env = getattr(opts, "true_dynamics", None)
        if env is not None:      # the simulator is the true dynamics of the rollout (mpc/lqr_step.py:223-225)
            assert opts.u_zero_I is None and opts.delta_u is None
            lo, hi = _bound(opts.u_lower), _bound(opts.u_upper)
            assert lo is None or isinstance(lo, float)
            nx, nu, cs, fdn, al = E.rollout(env.kind, _np(env.params).astype(np.float64), _np(x_init).astype(np.float64),
                                            _np(C).astype(np.float64), _np(c).astype(np.float64), o["K"].astype(np.float64),
                                            o["k"].astype(np.float64), _np(cur_x).astype(np.float64),
                                            _np(cur_u).astype(np.float64), lo, hi, opts.linesearch_decay,
                                            opts.max_linesearch_iter)
            o.update(new_x=nx, new_u=nu, costs=cs, full_du_norm=fdn, alpha_du_norm=fdn, alphas=al)
        res = {k: self._t(o[k], C) for k in ("new_x", "new_u", "costs", "old_costs", "full_du_norm",
                                             "alpha_du_norm", "alphas", "K", "k")}
        res["qp_iters"] = torch.full((B,), int(o["n_qp_iter"]), dtype=torch.int32)
        # the stand-in plays a kernel that TESTS C's symmetry unless told otherwise (MPC_ST_C_TESTED = 32, include/mpc_lqr.h);
        # `tests_c = False` plays the generic / lane-per-problem kernels, which never look
        tested = getattr(self, "tests_c", True) and not getattr(opts, "c_symmetric", False)
        res["status"] = torch.full((B,), 32 if tested else 0, dtype=torch.int32)
        if out_x is not None:          # (the device backend's kernel writes into the caller's buffers: mpc.shard's gather slots)
            out_x.copy_(res["new_x"]); out_u.copy_(res["new_u"])
            res["new_x"], res["new_u"] = out_x, out_u
        return res

    def plan_step(self, x_init, C, c, F, f, cur_x, cur_u, opts, impl=0, out_x=None, out_u=None, workspace=None):
        def run():
            self.calls.append("step:c_symmetric" if getattr(opts, "c_symmetric", False) else "step:c_unknown")
            r = self.lqr_step(x_init, C, c, F, f, cur_x, cur_u, opts)
            if out_x is not None:
                out_x.copy_(r["new_x"]); out_u.copy_(r["new_u"])
                r["new_x"], r["new_u"] = out_x, out_u
            return r
        return run

    def lqr_sweep(self, x_init, C, c, F, cur_x, cur_u, opts):
        self.calls.append("lqr_sweep")
        r = self.lqr_step(x_init, C, c, F, None, cur_x, cur_u, opts, want_gains=True)
        return {k: r[k] for k in ("K", "k", "old_costs", "qp_iters", "status")}

    def lqr_rollout(self, x_init, C, c, F, f, cur_x, cur_u, K, k, opts, old_costs=None):
        """lqr_forward given the gains -- the FULL step only (max_linesearch_iter = 1: what `reference_du_norm` asks for),
        mpc/lqr_step.py:181-241 in numpy."""
        assert opts.max_linesearch_iter == 1
        self.calls.append("lqr_rollout")
        T, B = C.shape[0], C.shape[1]
        Cn, cn, Kn, kn, xn, un = (_np(v).astype(np.float64) for v in (C, c, K, k, cur_x, cur_u))
        Fn = _np(F).astype(np.float64) if T > 1 else None
        fn = None if f is None or f.numel() == 0 else _np(f).astype(np.float64)
        lo, hi = _bound(opts.u_lower), _bound(opts.u_upper)
        zm = _np(opts.u_zero_I)
        x = _np(x_init).astype(np.float64)
        xs, us, cost = [], [], np.zeros(B)
        for t in range(T):
            nu = np.einsum("bij,bj->bi", Kn[t], x - xn[t]) + un[t] + kn[t]
            if zm is not None:
                nu = np.where(zm[t].astype(bool), 0.0, nu)
            if lo is not None:
                l = lo if isinstance(lo, float) else np.asarray(lo, np.float64)[t]
                h = hi if isinstance(hi, float) else np.asarray(hi, np.float64)[t]
                if opts.delta_u is not None:
                    l, h = np.maximum(l, un[t] - opts.delta_u), np.minimum(h, un[t] + opts.delta_u)
                nu = np.minimum(np.maximum(nu, l), h)
            tau = np.concatenate((x, nu), 1)
            cost += 0.5 * np.einsum("bi,bij,bj->b", tau, Cn[t], tau) + (tau * cn[t]).sum(1)
            xs.append(x); us.append(nu)
            if t < T - 1:
                x = np.einsum("bij,bj->bi", Fn[t], tau) + (0.0 if fn is None else fn[t])
        nx, nu_ = np.stack(xs), np.stack(us)
        dn = np.sqrt(((un - nu_) ** 2).sum((0, 2)))
        return dict(new_x=self._t(nx, C), new_u=self._t(nu_, C), costs=self._t(cost, C), full_du_norm=self._t(dn, C),
                    alpha_du_norm=self._t(dn, C), alphas=torch.ones(B, dtype=C.dtype))

    def du_norm_reference(self, u, new_u):
        """The reference's own expression, mpc/lqr_step.py:243-245."""
        self.calls.append("du_norm_reference")
        return (u - new_u).transpose(1, 2).contiguous().view(u.shape[1], -1).norm(2, 1)

    def kkt_backward(self, C, c, F, f, x_star, u_star, dl_dx, dl_du, opts, impl=0):
        self.calls.append("kkt_backward")
        o = O.kkt_backward(_np(C), _np(c), _np(F), _np(f), _np(x_star), _np(u_star),
                           _np(dl_dx.to(C.dtype)), _np(dl_du.to(C.dtype)),
                           _bound(opts.u_lower), _bound(opts.u_upper), lockstep=self.lockstep)
        return {k: (None if o[k] is None else self._t(o[k], C)) for k in ("dx_init", "dC", "dc", "dF", "df", "dx", "du")}

    def pnqp(self, H, q, lower, upper, x_init=None, n_iter=20, want_Hfree=True, want_lu=False):
        self.calls.append("pnqp")
        o = O.pnqp(_np(H), _np(q), _bound(lower), _bound(upper), _np(x_init), n_iter=n_iter, lockstep=self.lockstep)
        Hn = _np(H)
        If = o["If"].astype(bool)
        Hfree = np.where(If[:, :, None] & If[:, None, :], Hn, 0.0) + 1e-11 * np.eye(Hn.shape[1])
        LU = piv = None
        if want_lu:      # the checker's own factorisation (LAPACK through torch on the CPU)
            LU, piv = torch.linalg.lu_factor(self._t(Hfree, H))
        return dict(x=self._t(o["x"], H), If=torch.from_numpy(o["If"]), iters=torch.from_numpy(o["iters"]),
                    status=torch.from_numpy((1 - o["converged"]).astype(np.int32)), Hfree=self._t(Hfree, H),
                    LU=LU, pivots=piv)

    def traj_cost(self, x_init, u, F, f, C=None, c=None, want_x=True):
        self.calls.append("traj_cost")
        T, B, nc = u.shape
        Fn = _np(F) if T > 1 else np.zeros((0, B, x_init.shape[1], x_init.shape[1] + nc), _np(u).dtype)
        x, cost = O.traj_cost(_np(x_init), _np(u), Fn, _np(f), _np(C), _np(c))
        return (self._t(x, u) if want_x else None), (None if cost is None else self._t(cost, u))

    def env_traj_cost(self, x_init, u, env, C=None, c=None, want_x=True):
        self.calls.append("env_traj_cost")
        x = E.traj(env.kind, _np(x_init).astype(np.float64), _np(u).astype(np.float64), _np(env.params).astype(np.float64))
        cost = None
        if C is not None:
            cost = self._t(E.quad_cost(_np(C).astype(np.float64), _np(c).astype(np.float64), x, _np(u).astype(np.float64)), u)
        return (self._t(x, u) if want_x else None), cost

    def env_linearize(self, env, x, u, out_F=None, out_f=None):
        self.calls.append("env_linearize")
        F, f = E.linearize(env.kind, _np(x), _np(u), _np(env.params).astype(np.float64))
        F, f = self._t(F, x), self._t(f, x)
        if out_F is not None:
            out_F.copy_(F.view_as(out_F)); out_f.copy_(f.view_as(out_f))
            return out_F, out_f
        return F, f

    def select_best(self, first, eps, x, u, costs, du_norm, best, flags=None, status=None):
        self.calls.append("select_best")
        take = torch.ones_like(costs, dtype=torch.bool) if first else costs <= best["costs"] + eps
        best["x"][:, take] = x[:, take]
        best["u"][:, take] = u[:, take]
        best["costs"][take] = costs[take]
        best["full_du_norm"][take] = du_norm[take]
        # bit 1: C is NOT known to be symmetric -- MPC_ST_C_ASYMMETRIC (8) somewhere, or a status without MPC_ST_C_TESTED (32)
        asym = status is not None and bool((((status & 8) != 0) | ((status & 32) == 0)).any())
        any_improved = torch.tensor([int((not first) and bool(take.any())) | (2 if asym else 0)], dtype=torch.int32)
        if flags is not None:
            flags[0].copy_(any_improved); flags[1].copy_(du_norm.max().reshape(1))
            return flags
        return any_improved, du_norm.max().reshape(1)

    # -- NNDynamics through the oracle (oracle/env_oracle.py, MLP): what MPC._iterate_network drives on the device ----------
    @staticmethod
    def _mlp(net):
        return E.Mlp([_np(W).astype(np.float64) for W in net.weights], [_np(b).astype(np.float64) for b in net.biases],
                     net.activation, net.passthrough)

    def mlp_traj_cost(self, x_init, u, net, C=None, c=None):
        self.calls.append("mlp_traj_cost")
        x = E.traj(E.MLP, _np(x_init).astype(np.float64), _np(u).astype(np.float64), self._mlp(net))
        cost = None if C is None else self._t(E.quad_cost(_np(C).astype(np.float64), _np(c).astype(np.float64), x, _np(u).astype(np.float64)), u)
        return self._t(x, u), cost

    def plan_network_iteration(self, x_init, C, c, net, opts, nominals, scratch=None):
        """The stand-in of HipBackend.plan_network_iteration: linearise (MPC.linearize_dynamics, ANALYTIC), sweep, rollout
        through the network with the line search -- each piece the oracle's."""
        T, B = C.shape[0], C.shape[1]
        ns = x_init.shape[1]
        nc = C.shape[2] - ns
        mlp = self._mlp(net)
        outs = tuple(dict(new_x=nominals[1 - j][0], new_u=nominals[1 - j][1]) for j in (0, 1))
        vouched = []

        def run(j, stream=None):
            self.calls.append("network_iteration" + (":c_symmetric" if vouched else ""))
            cx, cu = (_np(t).astype(np.float64) for t in nominals[j])
            Fl, fl = E.linearize(E.MLP, cx[:-1].reshape(-1, ns), cu[:-1].reshape(-1, nc), mlp)
            o = O.lqr_step(_np(x_init), _np(C), _np(c), Fl.reshape(T - 1, B, ns, ns + nc), fl.reshape(T - 1, B, ns), cx, cu,
                           _bound(opts.u_lower), _bound(opts.u_upper), _np(opts.u_zero_I), opts.delta_u, opts.linesearch_decay,
                           opts.max_linesearch_iter, lockstep=self.lockstep, return_gains=True)
            nx, nu, cs, full, al, _tr, old = E.rollout_batched(E.MLP, mlp, _np(x_init).astype(np.float64), _np(C).astype(np.float64),
                                                               _np(c).astype(np.float64), o["K"], o["k"], cx, cu, _bound(opts.u_lower),
                                                               _bound(opts.u_upper), opts.linesearch_decay, opts.max_linesearch_iter,
                                                               delta_u=opts.delta_u, u_zero_I=_np(opts.u_zero_I))
            r = outs[j]
            r["new_x"].copy_(self._t(nx, C)); r["new_u"].copy_(self._t(nu, C))
            r.update(costs=self._t(cs, C), old_costs=self._t(old, C), full_du_norm=self._t(full, C), alpha_du_norm=self._t(full, C),
                     alphas=self._t(al, C), qp_iters=torch.full((B,), int(o["n_qp_iter"]), dtype=torch.int32),
                     status=torch.full((B,), 0 if vouched else 32, dtype=torch.int32))
            return r
        return run, outs, lambda: vouched.append(True)
