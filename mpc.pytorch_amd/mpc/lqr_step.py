"""`LQRStep(...)` -- one box-constrained LQR solve in delta space, as an autograd.Function.

Host-side mirror of the reference's mpc/lqr_step.py:22-409 (same factory signature, same return
tuples, same gradient convention).  What differs is where the work happens:

  forward   c_back + Riccati sweep + pnqp + line-searched rollout  -> ONE kernel launch
            (mpc_lqr_step; reference: ~4,000 ATen ops and several host syncs)
  backward  KKT system = one more LQR step on (C, -r, F) + a costate / outer-product kernel
            (mpc_lqr_kkt_prepare, mpc_lqr_step, mpc_lqr_kkt_grads)

Module-valued `true_dynamics` / `true_cost` keep the sweep on the kernel; the shipped simulators and
`mpc.dynamics.NNDynamics` (fp32) also roll out inside kernels, any other module is called timestep by
timestep in a loop of device ops.
"""
import os
import weakref
from collections import OrderedDict, namedtuple

import torch
from torch.autograd import Function

from . import _native
from ._native import StepOptions

LqrBackOut = namedtuple("lqrBackOut", "n_total_qp_iter")
LqrForOut = namedtuple("lqrForOut", "objs full_du_norm alpha_du_norm mean_alphas costs")


def _is_empty(f):
    return f is None or f.numel() == 0


def _same_storage(a, b):
    if a is None or b is None:
        return a is None and (b is None or b.numel() == 0)
    return a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.stride() == b.stride()


def _bound_at(v, t):
    return v if isinstance(v, (float, int)) else v[t]


def _rollout_pass(T, x_init, K, k, cur_x, cur_u, alpha, true_cost, true_dynamics, opts):
    """One pass of lqr_forward (mpc/lqr_step.py:181-241) with a module as dynamics and / or cost, every problem with its
    own step size alpha [B,1]: device ops only, nothing read back.  Returns new_x, new_u, cost [B], ||u - u'|| [B]."""
    from . import mpc as _mpc
    lin = isinstance(true_dynamics, _mpc.LinDx)
    quad = isinstance(true_cost, _mpc.QuadCost)
    xs, us, objs = [x_init], [], []
    dx = torch.zeros_like(x_init)
    for t in range(T):
        ut = cur_u[t]
        nu = torch.einsum("bij,bj->bi", K[t], dx) + ut + alpha * k[t]
        if opts.u_zero_I is not None:
            nu = nu.masked_fill(opts.u_zero_I[t].bool(), 0.0)
        if opts.u_lower is not None:
            lo, hi = _bound_at(opts.u_lower, t), _bound_at(opts.u_upper, t)
            if opts.delta_u is not None:          # (no host scalar becomes a device tensor here: that copy cannot be captured)
                lo = torch.maximum(ut - opts.delta_u, lo.to(ut.dtype)) if torch.is_tensor(lo) else (ut - opts.delta_u).clamp(min=float(lo))
                hi = torch.minimum(ut + opts.delta_u, hi.to(ut.dtype)) if torch.is_tensor(hi) else (ut + opts.delta_u).clamp(max=float(hi))
            if torch.is_tensor(lo):
                nu = torch.min(torch.max(nu, lo.to(nu.dtype)), hi.to(nu.dtype))
            else:
                nu = nu.clamp(float(lo), float(hi))
        us.append(nu)
        tau = torch.cat((xs[t], nu), 1)
        if t < T - 1:
            if lin:
                nx = torch.einsum("bij,bj->bi", true_dynamics.F[t].detach(), tau)
                if not _is_empty(true_dynamics.f):
                    nx = nx + true_dynamics.f[t].detach()
            else:
                nx = true_dynamics(xs[t], nu).detach()
            xs.append(nx)
            dx = nx - cur_x[t + 1]
        if quad:
            Ct, ct = true_cost.C[t].detach(), true_cost.c[t].detach()
            objs.append(0.5 * torch.einsum("bi,bij,bj->b", tau, Ct, tau) + (tau * ct).sum(1))
        else:
            objs.append(true_cost(tau).detach())
    new_x, new_u = torch.stack(xs), torch.stack(us)
    cost = torch.stack(objs).sum(0)
    du_norm = (cur_u - new_u).pow(2).sum((0, 2)).sqrt()
    return new_x, new_u, cost, du_norm


class _GraphedPass:
    """`_rollout_pass` for one (module, shapes, options) captured in a HIP graph: a pass is T x ~15 small launches, which
    the host issues slower than the device runs them; replayed, the pass costs one launch.  The gains, the nominal and the
    step sizes are copied into the graph's static buffers (they change every call); the cost's C, c, tensor bounds and the
    module's parameters are captured by address -- the cache key holds those addresses, an optimiser updating weights in
    place is seen.  Capturing runs the module three extra times (two warm-ups on a side stream + the capture itself)."""

    def __init__(self, T, x_init, K, k, cur_x, cur_u, true_cost, true_dynamics, opts):
        self.static = [t.clone() for t in (x_init, K, k, cur_x, cur_u)]
        self.alpha = torch.ones(x_init.shape[0], 1, dtype=x_init.dtype, device=x_init.device)
        run = lambda: _rollout_pass(T, *self.static, self.alpha, true_cost, true_dynamics, opts)
        side = torch.cuda.Stream(device=x_init.device)
        side.wait_stream(torch.cuda.current_stream(x_init.device))
        with torch.cuda.stream(side):
            for _ in range(2):
                run()
        torch.cuda.current_stream(x_init.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = run()

    def __call__(self, x_init, K, k, cur_x, cur_u, alpha):
        for dst, src in zip(self.static, (x_init, K, k, cur_x, cur_u)):
            dst.copy_(src)
        self.alpha.copy_(alpha)
        self.graph.replay()
        return self.out


_PASS_GRAPHS = OrderedDict()       # key -> (_GraphedPass | None when capture failed, weak refs to the modules)
_PASS_GRAPHS_MAX = 4


def _addr(v):
    return v.data_ptr() if torch.is_tensor(v) else v


def _graphed_pass(T, x_init, K, k, cur_x, cur_u, true_cost, true_dynamics, opts):
    """The cached graph for this rollout, or None.  OPT-IN: a replayed graph repeats the device work it recorded, so a
    module whose behaviour depends on Python-side state (a flag, a counter, `train()` / `eval()`) would silently keep its
    recorded behaviour.  A module declares itself safe with the attribute `hip_graph = True` (or MPC_ROLLOUT_GRAPH=1
    turns it on for all; MPC_NO_ROLLOUT_GRAPH=1 off for all).  A module that cannot be captured -- anything that
    synchronises, e.g. `.item()` -- is remembered and called eagerly."""
    from . import mpc as _mpc
    wanted = (os.environ.get("MPC_ROLLOUT_GRAPH") or getattr(true_dynamics, "hip_graph", False)
              or getattr(true_cost, "hip_graph", False))
    if (not wanted or not x_init.is_cuda or os.environ.get("MPC_NO_ROLLOUT_GRAPH")
            or torch.cuda.is_current_stream_capturing()):
        return None

    def ident(obj):
        if isinstance(obj, _mpc.QuadCost):
            return ("quad", obj.C.data_ptr(), obj.c.data_ptr(), tuple(obj.C.stride()))
        if isinstance(obj, _mpc.LinDx):
            return ("lin", obj.F.data_ptr(), _addr(obj.f) if not _is_empty(obj.f) else 0)
        return ("mod", id(obj))
    key = (ident(true_cost), ident(true_dynamics), T, tuple(K.shape), x_init.dtype, x_init.device.index,
           _addr(opts.u_lower), _addr(opts.u_upper), _addr(opts.u_zero_I), opts.delta_u)
    refs = tuple(weakref.ref(o) for o in (true_cost, true_dynamics) if isinstance(o, torch.nn.Module))
    hit = _PASS_GRAPHS.get(key)
    if hit is not None and all(r() is o for r, o in zip(hit[1], (o for o in (true_cost, true_dynamics) if isinstance(o, torch.nn.Module)))):
        _PASS_GRAPHS.move_to_end(key)
        return hit[0]
    try:
        g = _GraphedPass(T, x_init, K, k, cur_x, cur_u, true_cost, true_dynamics, opts)
    except Exception:                 # not capturable: remember, and run it eagerly from now on
        g = None
        torch.cuda.synchronize(x_init.device)
    _PASS_GRAPHS[key] = (g, refs)
    while len(_PASS_GRAPHS) > _PASS_GRAPHS_MAX:
        _PASS_GRAPHS.popitem(last=False)
    return g


def _module_rollout(n_state, n_ctrl, T, x_init, K, k, cur_x, cur_u, old_cost, true_cost, true_dynamics,
                    opts):
    """Rollout + per-problem line search when the dynamics or the cost is an nn.Module
    (reference mpc/lqr_step.py:164-261).  Device ops only; the per-element step size is a [B,1]
    column instead of the reference's B x B diag matrix (:192).  For a module that opts in (`hip_graph = True`) a pass
    is one replay of a HIP graph (`_GraphedPass`: a pass is launch-bound at small batches); the only read-back is the
    line search's own "did any cost get worse" (:176-179)."""
    B = x_init.shape[0]
    alpha = torch.ones(B, 1, dtype=x_init.dtype, device=x_init.device)
    full_du_norm = None
    with torch.no_grad():
        graphed = _graphed_pass(T, x_init, K, k, cur_x, cur_u, true_cost, true_dynamics, opts)
        for it in range(opts.max_linesearch_iter):
            if graphed is not None:
                new_x, new_u, cost, du_norm = graphed(x_init, K, k, cur_x, cur_u, alpha)
            else:
                new_x, new_u, cost, du_norm = _rollout_pass(T, x_init, K, k, cur_x, cur_u, alpha, true_cost,
                                                            true_dynamics, opts)
            if full_du_norm is None:
                full_du_norm = du_norm.clone()
            worse = cost > old_cost
            last = it + 1 >= opts.max_linesearch_iter
            if last or not bool(worse.any()):
                break
            alpha = torch.where(worse.unsqueeze(1), alpha * opts.linesearch_decay, alpha)
        if graphed is not None:       # the graph's output buffers are overwritten by the next replay
            new_x, new_u, cost, du_norm = new_x.clone(), new_u.clone(), cost.clone(), du_norm.clone()
    return new_x, new_u, cost, full_du_norm, du_norm, alpha.squeeze(1)


class _AsyncHostScalar(torch.Tensor):
    """A 1-element CPU tensor whose value arrives by an asynchronous device->host copy; the first torch operation
    that touches it (float(), .item(), printing, arithmetic, .numpy(), torch.stack([...]), torch.add(x, other=...))
    first waits for the EVENT recorded behind the copy -- not for the whole stream -- and moves the value out of its
    pinned landing slot into storage of its own (the slots are a ring that later solves reuse).  What `LQRStep(...)`
    returns as `n_total_qp_iter`: the reference hands back a CPU float tensor there (mpc/lqr_step.py:308) and pays a
    device synchronisation for it in every forward; here nothing waits unless somebody looks."""

    @staticmethod
    def __new__(cls, host, event):
        t = torch.Tensor._make_subclass(cls, host)
        t._event = event
        return t

    def _settle(self):
        ev = getattr(self, "_event", None)
        if ev is not None:
            self._event = None
            ev.synchronize()
            with torch._C.DisableTorchFunctionSubclass():
                own = torch.Tensor.clone(self)
                torch.Tensor.set_(self, own.untyped_storage(), 0, own.shape, own.stride())
            cb = getattr(self, "_on_settle", None)          # (mpc.pnqp: the reference's convergence warning rides on the same read)
            if cb is not None:
                self._on_settle = None
                cb()

    def __index__(self):
        # (pnqp's 4th return value is the Python int `i` in the reference, mpc/pnqp.py:59, 82 -- range(i), seq[i] and
        # friends go through __index__, which torch refuses for a float tensor)
        self._settle()
        with torch._C.DisableTorchFunctionSubclass():
            return int(torch.Tensor.item(self))

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        from torch.utils._pytree import tree_flatten
        for a in tree_flatten((args, kwargs or {}))[0]:          # nested lists / tuples / keyword arguments too
            if isinstance(a, cls):
                a._settle()
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))


_PINNED_RING = {}      # device index -> [pinned float32 tensor of _RING slots, next slot, the event guarding each slot]
_RING = 256


def _host_scalar_async(dev_scalar):
    """int32 / float device tensor [1] -> CPU float tensor [1], without synchronising (see _AsyncHostScalar).
    The pinned landing slots are a ring per device; a value leaves its slot the first time it is looked at, and a slot
    is only written again once the copy that last used it has completed (its event is waited for: 256 solves later,
    i.e. never in practice)."""
    if not dev_scalar.is_cuda:
        return _AsyncHostScalar(dev_scalar.to(torch.float32).reshape(1).cpu().clone(), None)     # (settled: nothing to wait for)
    key = dev_scalar.device.index
    ring = _PINNED_RING.get(key)
    if ring is None:
        ring = _PINNED_RING[key] = [torch.zeros(_RING, dtype=torch.float32).pin_memory(), 0, [None] * _RING]
    i = ring[1]
    ring[1] = (i + 1) % _RING
    prev = ring[2][i]
    if prev is not None:
        prev[0].synchronize()
        holder = prev[1]()
        if holder is not None:
            holder._settle()              # an unread value still living in the slot: move it out before the slot is reused
    slot = ring[0][i:i + 1]
    slot.copy_(dev_scalar.to(torch.float32).reshape(1), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev_scalar.device))
    out = _AsyncHostScalar(slot, ev)
    ring[2][i] = (ev, weakref.ref(out))
    return out


class _StepConfig:
    """What one LQRStep(...) call fixes for its autograd node (closure state of the reference's factory)."""
    __slots__ = ("solve", "no_op_forward", "delta_space", "current_x", "current_u", "u_lower", "u_upper", "c_symmetric")


class _LQRStepFn(Function):
    """One autograd node type for every LQRStep.  (The reference defines a new Function class inside each
    LQRStep(...) call, mpc/lqr_step.py:275; a class object is only ever freed by the cyclic garbage collector,
    which keeps every solve's tensors alive until it happens to run.)"""

    @staticmethod
    def forward(ctx, cfg, x_init, C, c, F, f=None):
        if f is None:
            f = torch.empty(0)
        # Only the bounds go on ctx.  (The reference also parks current_x / current_u there, :281, 300-301: in
        # the no-op forward those ARE the outputs, and output -> grad_fn -> ctx -> output is a cycle the
        # garbage collector cannot see -- every differentiated solve would pin its tensors forever.)
        ctx.u_lower, ctx.u_upper = cfg.u_lower, cfg.u_upper
        ctx.c_symmetric = cfg.c_symmetric
        if cfg.no_op_forward:
            ctx.save_for_backward(x_init, C, c, F, f, cfg.current_x, cfg.current_u)
            return cfg.current_x, cfg.current_u
        if not cfg.delta_space:
            assert False      # unimplemented upstream too (mpc/lqr_step.py:297-298)
        assert cfg.current_x is not None
        assert cfg.current_u is not None
        new_x, new_u, qp_iters, costs, full_du_norm, alphas = cfg.solve(x_init, C, c, F, f)
        ctx.save_for_backward(x_init, C, c, F, f, new_x, new_u)
        # n_total_qp_iter stays on the device here; LQRStep's wrapper turns it into the CPU float tensor of the
        # reference (mpc/lqr_step.py:308) by an asynchronous copy -- no host synchronisation in a forward.
        # Value: max over the problems of sum_t (1 + that problem's pnqp iterations).  The reference's loops are
        # batch-global, so it reports sum_t (1 + max over the batch) >= this; they agree for n_batch = 1.  Only the
        # "total_qp_iters" log column and this third return value see it.
        n_qp = qp_iters.max().reshape(1) if cfg.u_lower is not None else qp_iters[:1]     # unbounded: zeros, no kernel
        ctx.mark_non_differentiable(n_qp)
        return new_x, new_u, n_qp, costs, full_du_norm, alphas.mean()

    @staticmethod
    def backward(ctx, dl_dx, dl_du, *unused):
        x_init, C, c, F, f, new_x, new_u = ctx.saved_tensors
        if dl_dx is None:
            dl_dx = torch.zeros_like(new_x)
        if dl_du is None:
            dl_du = torch.zeros_like(new_u)
        g = _native.backend().kkt_backward(
            C, c, F, None if _is_empty(f) else f, new_x, new_u, dl_dx, dl_du,
            StepOptions(u_lower=ctx.u_lower, u_upper=ctx.u_upper, c_symmetric=ctx.c_symmetric))
        df = g["df"] if g["df"] is not None else torch.Tensor()
        return None, g["dx_init"], g["dC"], g["dc"], g["dF"], df


def LQRStep(n_state,
            n_ctrl,
            T,
            u_lower=None,
            u_upper=None,
            u_zero_I=None,
            delta_u=None,
            linesearch_decay=0.2,
            max_linesearch_iter=10,
            true_cost=None,
            true_dynamics=None,
            delta_space=True,
            current_x=None,
            current_u=None,
            verbose=0,
            back_eps=1e-3,
            no_op_forward=False,
            c_symmetric=False,
            reference_du_norm=False):
    """A single step of the box-constrained iLQR solver.

    Required: n_state, n_ctrl, T.  The returned callable takes (x_init [B,ns], C [T,B,n,n],
    c [T,B,n], F [T-1,B,ns,n], f [T-1,B,ns] or an EMPTY tensor) and returns
    (new_x, new_u, n_total_qp_iter, costs, full_du_norm, mean_alphas), or (current_x, current_u)
    when `no_op_forward` (used to attach the backward to an already-converged trajectory).

    u_lower / u_upper: python floats or [T, n_batch, n_ctrl] tensors.

    c_symmetric (not in the reference): the caller vouches that every C_t is symmetric, which lets the fused kernels
    skip their symmetry test (MPC_OPT_C_SYMMETRIC).  Left False, a C that is not symmetric is detected on the device
    and solved the reference's way (it uses C as given, mpc/lqr_step.py:68, 294) on the generic kernels.

    reference_du_norm (not in the reference): `full_du_norm` as the reference computes it for n_batch > 1 -- (u - new_u)
    .transpose(1, 2).contiguous().view(n_batch, -1).norm(2, 1), mpc/lqr_step.py:243-245: the transpose in front of the reshape
    makes entry r the norm of T n_ctrl consecutive elements of the [T, n_ctrl, n_batch] array, which mixes the problems of the
    batch.  Off (default), entry b is problem b's own norm (= the reference called with n_batch = 1).  On, one more rollout (the
    full step, alpha = 1, for every problem: mpc_lqr_rollout) and mpc_du_norm_reference reproduce the reference's vector.
    """
    opts = StepOptions(u_lower=u_lower, u_upper=u_upper, u_zero_I=u_zero_I, delta_u=delta_u,
                       linesearch_decay=linesearch_decay, max_linesearch_iter=max_linesearch_iter,
                       c_symmetric=c_symmetric)

    def solve(x_init, C, c, F, f):
        from . import mpc as _mpc
        be = _native.backend()
        f_in = None if _is_empty(f) else f
        lin = true_dynamics is None or isinstance(true_dynamics, _mpc.LinDx)
        quad = true_cost is None or isinstance(true_cost, _mpc.QuadCost)
        # Currently unimplemented in the reference as well (mpc/lqr_step.py:195):
        assert not ((delta_u is not None) and (u_lower is None))
        B = C.shape[1]
        ref_norm = bool(reference_du_norm) and B > 1
        sim = hasattr(true_dynamics, "native_env") and not ref_norm      # (reference_du_norm: a simulator goes the module's way)
        if (lin or sim) and quad:
            rp = None
            tC, tc = (C, c) if true_cost is None else (true_cost.C, true_cost.c)
            tF, tf = (F, f_in) if (sim or true_dynamics is None) else (true_dynamics.F, true_dynamics.f)
            if not (_same_storage(tC, C) and _same_storage(tc, c) and _same_storage(tF, F)
                    and _same_storage(tf, f_in)):
                rp = (tC, tc, tF, tf)     # true cost / dynamics differ from the quadratic model
            o = opts
            if sim:                        # a shipped simulator rolls out inside the kernel (:223-225)
                o = StepOptions(u_lower=u_lower, u_upper=u_upper, u_zero_I=u_zero_I, delta_u=delta_u,
                                linesearch_decay=linesearch_decay, max_linesearch_iter=max_linesearch_iter,
                                true_dynamics=true_dynamics.native_env(), c_symmetric=c_symmetric)
            r = be.lqr_step(x_init, C, c, F, f_in, current_x, current_u, o, rollout_problem=rp, want_gains=ref_norm)
            fdn = r["full_du_norm"]
            if ref_norm:
                # the controls of the FULL step for every problem (the accepted trajectory of a problem that backtracked is
                # another one): lqr_forward once more with max_linesearch_iter = 1, then the reference's mixed-up norm
                o1 = StepOptions(u_lower=u_lower, u_upper=u_upper, u_zero_I=u_zero_I, delta_u=delta_u,
                                 linesearch_decay=linesearch_decay, max_linesearch_iter=1, c_symmetric=c_symmetric)
                full = be.lqr_rollout(x_init.detach(), tC.detach(), tc.detach(), tF.detach(), None if tf is None else tf.detach(),
                                      current_x.detach(), current_u.detach(), r["K"], r["k"], o1, old_costs=r["old_costs"])
                fdn = be.du_norm_reference(current_u.detach(), full["new_u"])
            return r["new_x"], r["new_u"], r["qp_iters"], r["costs"], fdn, r["alphas"]
        from . import util as _util
        net = true_dynamics.native_net(C) if quad and hasattr(true_dynamics, "native_net") and not ref_norm else None
        if net is not None:
            # NNDynamics (:223-225): the sweep on the fastest kernel for this shape (MPC_OPT_SWEEP_ONLY), then the
            # line-searched rollout through the network in one kernel
            cx, cu = current_x.detach(), current_u.detach()
            r = be.lqr_sweep(x_init.detach(), C.detach(), c.detach(), F.detach(), cx, cu, opts)
            tC, tc = (C, c) if true_cost is None else (true_cost.C, true_cost.c)
            old_cost = r["old_costs"]
            if not (_same_storage(tC, C) and _same_storage(tc, c)):
                old_cost = _util.get_cost(T, cu, true_cost, true_dynamics, x=cx)
            rr = be.mlp_rollout(x_init.detach(), tC.detach(), tc.detach(), r["K"], r["k"], cx, cu, old_cost, opts, net)
            return rr["new_x"], rr["new_u"], r["qp_iters"], rr["costs"], rr["full_du_norm"], rr["alphas"]
        sw = be.lqr_sweep(x_init.detach(), C, c, F, current_x.detach(), current_u.detach(), opts)
        old_cost = _util.get_cost(T, current_u.detach(), true_cost, true_dynamics, x=current_x.detach())
        nx, nu, cost, fdn, _, alphas = _module_rollout(
            n_state, n_ctrl, T, x_init.detach(), sw["K"], sw["k"], current_x.detach(), current_u.detach(),
            old_cost, true_cost, true_dynamics, opts)
        if ref_norm:
            with torch.no_grad():
                one = torch.ones(B, 1, dtype=x_init.dtype, device=x_init.device)
                full_u = _rollout_pass(T, x_init.detach(), sw["K"], sw["k"], current_x.detach(), current_u.detach(), one,
                                       true_cost, true_dynamics, opts)[1]
            fdn = be.du_norm_reference(current_u.detach(), full_u)
        return nx, nu, sw["qp_iters"], cost, fdn, alphas

    cfg = _StepConfig()
    cfg.solve, cfg.no_op_forward, cfg.delta_space = solve, no_op_forward, delta_space
    cfg.current_x, cfg.current_u = current_x, current_u
    cfg.u_lower, cfg.u_upper = u_lower, u_upper
    cfg.c_symmetric = bool(c_symmetric)

    def apply(x_init, C, c, F, f=None):
        out = _LQRStepFn.apply(cfg, x_init, C, c, F, f)
        if no_op_forward:
            return out
        n_qp = _host_scalar_async(out[2]) if u_lower is not None else torch.zeros(1)
        return out[:2] + (n_qp,) + out[3:]
    return apply
