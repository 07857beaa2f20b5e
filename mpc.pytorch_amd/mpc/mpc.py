"""`MPC` -- the differentiable box-constrained iLQR driver, MI355X build.

API mirror of the reference's mpc/mpc.py (`MPC`, `QuadCost`, `LinDx`, `GradMethods`): same
constructor keywords, same `(x, u, costs)` return, same convergence / best-iterate rules
(mpc/mpc.py:245-337).  The per-iteration work is the `mpc_lqr_step` kernel; the bookkeeping the
reference does with a Python loop over the batch (best-iterate copy, `max(full_du_norm)`,
mpc/mpc.py:279-285, 299) is the `mpc_select_best` kernel, so one iLQR iteration costs two launches
and ONE small device->host read.
"""
import collections
from collections import namedtuple
from enum import Enum

import torch
from torch.nn import Module

from . import _native, util
from ._native import StepOptions
from .lqr_step import LQRStep

QuadCost = namedtuple("QuadCost", "C c")
LinDx = namedtuple("LinDx", "F f")
QuadCost.__new__.__defaults__ = (None,) * len(QuadCost._fields)
LinDx.__new__.__defaults__ = (None,) * len(LinDx._fields)


class GradMethods(Enum):
    AUTO_DIFF = 1
    FINITE_DIFF = 2
    ANALYTIC = 3
    ANALYTIC_CHECK = 4


def _any_requires_grad(obj):
    """Does a QuadCost / LinDx / module carry a tensor that asks for a gradient?  Modules are searched
    through parameters, buffers and plain tensor attributes (the shipped simulators keep `params` as
    one), recursively over sub-modules."""
    if isinstance(obj, (QuadCost, LinDx)):
        return any(torch.is_tensor(t) and t.requires_grad for t in obj)
    if isinstance(obj, Module):
        for m in obj.modules():
            for v in list(m.parameters(recurse=False)) + list(vars(m).values()):
                if torch.is_tensor(v) and v.requires_grad:
                    return True
                if isinstance(v, (list, tuple)) and any(torch.is_tensor(t) and t.requires_grad for t in v):
                    return True
        return False
    return False


_PINNED = collections.OrderedDict()     # (device, stream, host thread) -> [pinned block, its int32 view, tag counter]; LRU, see _FlagReader
_PINNED_MAX = 64                         # a thread-per-request server would otherwise page-lock a block per thread for ever


class _FlagReader:
    """The two words the driver needs from the device per iteration (any problem improved, max ||du||).
    With the HIP backend the select kernel stores them in page-locked host memory itself, followed by a tag; `wait()` polls
    that tag -- no device-to-host copy and no event (an event's system-scope release held the next kernel back by 6 us per
    iteration).  Other backends (the test stand-ins): an asynchronous copy and an event."""

    _POLL_S = 0.005

    def __init__(self, device, dtype, be=None):
        # int32 flag at byte 0, the tag at byte 4, the maximum (float32 / float64) at byte 8 of one block
        self.cuda = device.type == "cuda"
        self._host_device = device
        self.direct = self.cuda and getattr(be, "writes_host_flags", False)
        if self.direct:
            self.device_flags = be.select_flags(device, dtype)
        else:
            self._dev = torch.empty(16, dtype=torch.uint8, device=device)
            self.device_flags = (self._dev[0:4].view(torch.int32), self._dev[8:8 + torch.empty(0, dtype=dtype).element_size()].view(dtype))
        if self.cuda:
            # page-locking memory costs milliseconds: one block per device, stream AND host thread, kept for the life of the
            # process (two solves on one stream from two threads would otherwise overwrite each other's tag and result words:
            # ADVICE r03; within a thread solves are sequential, the tag counter tells their select calls apart)
            import threading
            key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream, threading.get_ident())
            slot = _PINNED.get(key)
            if slot is None:
                blk = torch.zeros(16, dtype=torch.uint8).pin_memory()
                slot = _PINNED[key] = [blk, blk.numpy().view("int32"), 0]
                while len(_PINNED) > _PINNED_MAX:          # least recently used out; a reader still holding its slot keeps it alive
                    _PINNED.popitem(last=False)
            else:
                _PINNED.move_to_end(key)
            self._slot = slot
            self._host = self._slot[0]
            self.host = (self._host[0:4].view(torch.int32), self._host[8:8 + self.device_flags[1].element_size()].view(dtype))
            if not self.direct:
                self.event = torch.cuda.Event()

    def select_kw(self):
        """keyword arguments of the select call this reader will wait for"""
        if not self.direct:
            return {}
        self._slot[2] = (self._slot[2] % 0x7fffffff) + 1          # a tag no earlier call on this block has used (never 0)
        return dict(host=self._host, tag=self._slot[2])

    def start(self):
        if self.cuda and not self.direct:
            self._host.copy_(self._dev, non_blocking=True)
            self.event.record()

    def wait(self):
        """-> (flag bits: 1 = some problem improved, 2 = some C is not symmetric; max ||du||)"""
        if self.direct:
            words, tag = self._slot[1], self._slot[2]
            if words[1] != tag:
                import time
                t0 = time.monotonic()
                while words[1] != tag:
                    if time.monotonic() - t0 > self._POLL_S:
                        # not there after a few milliseconds: a long queue in front of the select kernel, or host memory
                        # the device's stores reach only when the kernel retires -- wait for the stream (which also
                        # surfaces a launch failure as the error it is); the words are there afterwards or never
                        torch.cuda.current_stream(self._host_device).synchronize()
                        if words[1] != tag:
                            raise RuntimeError("mpc_select_best did not report (tag %d, host word %d)" % (tag, int(words[1])))
            return int(self.host[0][0]), float(self.host[1][0])
        if self.cuda:
            self.event.synchronize()
            return int(self.host[0][0]), float(self.host[1][0])
        return int(self.device_flags[0][0]), float(self.device_flags[1][0])


class SlewRateCost(Module):
    """A module cost plus the slew-rate quadratic on the augmented variable (u_prev, x, u)
    (reference mpc/mpc.py:36-56)."""

    def __init__(self, cost, slew_C, n_state, n_ctrl):
        super().__init__()
        self.cost, self.slew_C = cost, slew_C
        self.n_state, self.n_ctrl = n_state, n_ctrl

    def forward(self, tau):
        return self.cost(tau[:, self.n_ctrl:]) + 0.5 * util.bquad(tau, self.slew_C[0])   # time-invariant

    def grad_input(self, x, u):
        raise NotImplementedError("Implement grad_input")


class UnconvergedError(AssertionError):
    """Raised where the reference does a bare `assert False` (mpc/mpc.py:321-324): some problem did
    not reach a fixed point and exit_unconverged=True."""


class MPC(Module):
    """A differentiable box-constrained iLQR solver.

        min_{tau={x,u}} sum_t 0.5 tau_t^T C_t tau_t + c_t^T tau_t
                        s.t. x_{t+1} = f(x_t, u_t),  x_0 = x_init,  u_lower <= u <= u_upper

    Constructor arguments are those of the reference (mpc/mpc.py:123-144): n_state, n_ctrl, T,
    u_lower/u_upper (floats or [T,B,n_ctrl]), u_zero_I, u_init, lqr_iter, grad_method, delta_u,
    verbose, eps, back_eps, n_batch, linesearch_decay, max_linesearch_iter, exit_unconverged,
    detach_unconverged, backprop, slew_rate_penalty, prev_ctrl, not_improved_lim, best_cost_eps.
    """

    def __init__(self, n_state, n_ctrl, T, u_lower=None, u_upper=None, u_zero_I=None, u_init=None,
                 lqr_iter=10, grad_method=GradMethods.ANALYTIC, delta_u=None, verbose=0, eps=1e-7,
                 back_eps=1e-7, n_batch=None, linesearch_decay=0.2, max_linesearch_iter=10,
                 exit_unconverged=True, detach_unconverged=True, backprop=True, slew_rate_penalty=None,
                 prev_ctrl=None, not_improved_lim=5, best_cost_eps=1e-4, reference_du_norm=False):
        super().__init__()
        assert (u_lower is None) == (u_upper is None)
        assert max_linesearch_iter > 0
        self.n_state, self.n_ctrl, self.T = n_state, n_ctrl, T
        self.u_lower = u_lower if isinstance(u_lower, float) else util.detach_maybe(u_lower)
        self.u_upper = u_upper if isinstance(u_upper, float) else util.detach_maybe(u_upper)
        self.u_zero_I = util.detach_maybe(u_zero_I)
        self.u_init = util.detach_maybe(u_init)
        self.lqr_iter = lqr_iter
        self.grad_method = grad_method
        self.delta_u = delta_u
        self.verbose = verbose
        self.eps = eps
        self.back_eps = back_eps
        self.n_batch = n_batch
        self.linesearch_decay = linesearch_decay
        self.max_linesearch_iter = max_linesearch_iter
        self.exit_unconverged = exit_unconverged
        self.detach_unconverged = detach_unconverged
        self.backprop = backprop
        self.not_improved_lim = not_improved_lim
        self.best_cost_eps = best_cost_eps
        self.slew_rate_penalty = slew_rate_penalty
        # (not in the reference) True: `full_du_norm` as the reference computes it for n_batch > 1 -- a transpose in front of the
        # reshape mixes the problems of a batch (mpc/lqr_step.py:243-245) -- so that the eps exit (:299) and the detach mask
        # (:321-334) are the reference's.  Default: each problem's own norm.  Costs one more rollout per iteration
        # and takes the general loop (no pre-bound device-side iterations).
        self.reference_du_norm = bool(reference_du_norm)
        self.flag_reducer = None     # set by mpc.shard for lock-step sharded solves
        self.prev_ctrl = prev_ctrl

    # ------------------------------------------------------------------------------------------
    def _expand_cost(self, cost, n_batch):
        """[n,n] / [T,n,n] costs are broadcast to [T,B,n,n] as stride-0 views (never copied: the
        kernels read them through their strides, so a shared cost matrix stays L2-resident)."""
        C, c = cost
        n = self.n_state + self.n_ctrl
        if C.ndimension() == 2:
            C = C.unsqueeze(0).unsqueeze(0).expand(self.T, n_batch, n, n)
        elif C.ndimension() == 3:
            C = C.unsqueeze(1).expand(self.T, n_batch, n, n)
        if c.ndimension() == 1:
            c = c.unsqueeze(0).unsqueeze(0).expand(self.T, n_batch, n)
        elif c.ndimension() == 2:
            c = c.unsqueeze(1).expand(self.T, n_batch, n)
        if C.ndimension() != 4 or c.ndimension() != 3:
            raise ValueError("MPC Error: Unexpected QuadCost shape.")
        return QuadCost(C, c)

    def forward(self, x_init, cost, dx):
        assert isinstance(cost, (QuadCost, Module)) or callable(cost)
        assert isinstance(dx, (LinDx, Module)) or callable(dx)
        if self.n_batch is not None:
            n_batch = self.n_batch
        elif isinstance(cost, QuadCost) and cost.C.ndimension() == 4:
            n_batch = cost.C.size(1)
        else:
            raise ValueError("MPC Error: Could not infer batch size, pass in as n_batch")
        if isinstance(cost, QuadCost):
            cost = self._expand_cost(cost, n_batch)
        assert x_init.ndimension() == 2 and x_init.size(0) == n_batch

        T, ns, nc = self.T, self.n_state, self.n_ctrl
        if self.u_init is None:
            u = torch.zeros(T, n_batch, nc, dtype=x_init.dtype, device=x_init.device)
        else:
            u = self.u_init
            if u.ndimension() == 2:
                u = u.unsqueeze(1).expand(T, n_batch, -1).clone()
            u = u.to(dtype=x_init.dtype, device=x_init.device)

        if self.verbose > 0:
            print("Initial mean(cost): {:.4e}".format(
                util.get_cost(T, u, cost, dx, x_init=x_init).mean().item()))

        ref_norm = self.reference_du_norm and n_batch > 1
        fast = (isinstance(cost, QuadCost) and isinstance(dx, LinDx) and self.slew_rate_penalty is None and not ref_norm)
        # a shipped simulator (mpc.env_dx): closed-form linearisation kernel + the simulator inside the
        # rollout kernel; FINITE_DIFF keeps the reference's central differences
        sim = None
        if (isinstance(cost, QuadCost) and hasattr(dx, "native_env") and self.slew_rate_penalty is None
                and self.grad_method in (GradMethods.ANALYTIC, GradMethods.AUTO_DIFF) and T > 1 and not ref_norm):
            sim = dx.native_env()
        be = _native.backend()
        # NNDynamics the kernels take (fp32 on the device, <= 4 layers, n_state <= 16), ANALYTIC linearisation: the whole
        # iteration is three pre-bound C calls (round 5; the general loop below pays allocations, struct rebuilds, an autograd
        # node and a device synchronisation per iteration: 1.1 ms an iteration for 0.52 ms of kernels at the headline shape)
        net = None
        if (sim is None and not fast and isinstance(cost, QuadCost) and self.slew_rate_penalty is None and T > 1
                and self.grad_method == GradMethods.ANALYTIC and hasattr(dx, "native_net")
                and hasattr(be, "plan_network_iteration") and not ref_norm):
            net = dx.native_net(x_init)
            if net is not None and net.activation == "elu":      # (no grad_input in the reference: the module refuses, mpc/dynamics.py:113-114)
                net = None
        if fast or sim is not None:
            best = self._iterate_planned(be, x_init, u, cost, dx, sim, n_batch)
        elif net is not None:
            best = self._iterate_network(be, x_init, u, cost, dx, net, n_batch)
        else:
            best = self._iterate_general(be, x_init, u, cost, dx)

        x, u = best["x"], best["u"]
        full_du_norm = best["full_du_norm"]
        if not self._wants_graph(x_init, cost, dx):
            # nothing upstream asks for a gradient: the differentiable re-linearisation and the no-op
            # step of mpc/mpc.py:308-319 would only build a graph nobody can reach
            self._check_converged(full_du_norm)
            return (x, u, best["costs"])
        if isinstance(dx, LinDx):
            F, f = dx.F, dx.f
        elif sim is not None and not _any_requires_grad(dx):
            # the simulator's parameters are constants here: closed-form kernel instead of autograd
            Fl, fl = be.env_linearize(sim, x[:-1].reshape(-1, ns), u[:-1].reshape(-1, nc))
            F, f = Fl.view(T - 1, n_batch, ns, ns + nc), fl.view(T - 1, n_batch, ns)
        else:
            F, f = self.linearize_dynamics(x, u, dx, diff=True)
        if isinstance(cost, QuadCost):
            C, c = cost.C, cost.c
        else:
            C, c, _ = self.approximate_cost(x, u, cost, diff=True)

        # attach the KKT backward at the best iterate (no compute), mpc/mpc.py:318-319
        x, u = self.solve_lqr_subproblem(x_init, C, c, F, f, cost, dx, x, u, no_op_forward=True)

        if self._check_converged(full_du_norm):
            keep = (full_du_norm < self.eps).to(x.dtype).view(1, -1, 1)
            x = x * keep + x.detach() * (1. - keep)
            u = u * keep + u.detach() * (1. - keep)
        return (x, u, best["costs"])

    def _iterate_planned(self, be, x_init, u, cost, dx, sim, n_batch):
        """The iLQR loop (mpc/mpc.py:245-306) when the whole iteration lives on the device: QuadCost with
        LinDx or a shipped simulator.  Inner iterations are never differentiated (the reference detaches
        them too).  Two pre-bound step plans ping-pong the nominal between two buffers, so an iteration is
        one C call (a simulator is linearised inside that call) -- no allocation, no autograd node.
        The states of a rollout ARE get_traj of its controls (:251 recomputes them).
        The convergence flags of iteration i are read back while iteration i+1 already runs: the next
        step is launched speculatively and simply not used if the flags say stop."""
        T, ns, nc = self.T, self.n_state, self.n_ctrl
        xi = util.detach_maybe(x_init)
        # the ping-pong plans WRITE into `ua`: it must be this solve's own buffer, never the caller's u_init
        # (detach / to / contiguous all return the same storage for a contiguous tensor)
        ua = util.detach_maybe(u).contiguous()
        if self.u_init is not None and ua.untyped_storage().data_ptr() == self.u_init.untyped_storage().data_ptr():
            ua = ua.clone()
        opts = self._step_options()
        if sim is not None:
            xa, _ = be.env_traj_cost(xi, ua, sim)                         # util.get_traj, :251
            # linearize_dynamics (:490-549) happens inside the step kernel: no F, f round trip through memory
            sim.linearize = True
            F = f = None
            opts.true_dynamics = sim
        else:
            xa = util.get_traj(T, ua, x_init=xi, dynamics=dx).contiguous()
            F, f = dx.F, dx.f
            # every nominal of this loop obeys (F, f) by construction: get_traj above, then each step's own rollout
            opts.nominal_on_dynamics = True
        xb, ub = torch.empty_like(xa), torch.empty_like(ua)
        pa = be.plan_step(xi, cost.C, cost.c, F, f, xa, ua, opts, out_x=xb, out_u=ub)
        r = pa()                  # the first step is on its way before anything else of the loop is set up (host time hidden)
        variant = getattr(be, "plan_variant", None)      # (the test backends build every plan from scratch)
        if variant is not None:
            pb = variant(pa, cur_x=xb, cur_u=ub, out_x=xa, out_u=ua)
        else:
            pb = be.plan_step(xi, cost.C, cost.c, F, f, xb, ub, opts, out_x=xa, out_u=ua)
        plans = (pa, pb)
        # C does not change during the solve: once the first step has reported that it is symmetric (no
        # MPC_ST_C_ASYMMETRIC in its status, read back with the convergence flags), the remaining steps run with the
        # promise MPC_OPT_C_SYMMETRIC -- no symmetry test in the kernel, no gated second launch behind it
        self._c_symmetric = False
        sym_plans = None

        # the raw stream handle, looked up once (torch.cuda.current_stream costs 4 us a call, two calls an iteration)
        stream = torch.cuda.current_stream(xa.device).cuda_stream if xa.is_cuda and variant is not None else None

        def launch(i):
            plan = (sym_plans if sym_plans is not None else plans)[i % 2]
            return plan() if stream is None else plan(stream)

        def on_symmetric():
            # (a simulator solve runs on the lane-per-problem kernel, which keeps Q and V general and has no test to
            #  skip: building two more plans would only cost host time in a loop that is launch-latency bound)
            nonlocal sym_plans
            if self.lqr_iter > 2 and sim is None:
                import copy
                so = copy.copy(opts)
                so.c_symmetric = True
                if variant is not None:          # the same structs with one more option bit: no walk, no allocation
                    sym_plans = (variant(pa, opts=so), variant(pb, opts=so))
                else:
                    sym_plans = (be.plan_step(xi, cost.C, cost.c, F, f, xa, ua, so, out_x=xb, out_u=ub),
                                 be.plan_step(xi, cost.C, cost.c, F, f, xb, ub, so, out_x=xa, out_u=ua))
        return self._drive(be, launch, (pa.outputs, pb.outputs) if variant is not None else None, r, xa, ua, n_batch, stream, on_symmetric)

    def _drive(self, be, launch, outputs, r, xa, ua, n_batch, stream, on_symmetric):
        """The loop of mpc/mpc.py:245-306 around pre-bound iterations: launch(i) enqueues iteration i and returns its output
        dict (`outputs`: the two dicts the iterations alternate between, for the pre-bound select call; None = the test
        backends' general entry), `r` = the outputs of iteration 0, already on its way.  Best-iterate tracking on the device
        (:271-285); the convergence flags of iteration i are read back while iteration i+1 already runs."""
        best = dict(x=torch.empty_like(xa), u=torch.empty_like(ua),
                    costs=torch.empty(n_batch, dtype=xa.dtype, device=xa.device),
                    full_du_norm=torch.empty(n_batch, dtype=xa.dtype, device=xa.device))
        reader = _FlagReader(xa.device, xa.dtype, be)
        # (the HIP backend binds the select call's arguments once per solve; the test stand-ins take the general entry)
        sel = None
        if reader.direct and hasattr(be, "plan_select") and outputs is not None:
            sel = be.plan_select(self.best_cost_eps, outputs, best, reader.device_flags, host=reader._host)
        n_not_improved, i = 0, 0
        while True:
            # best-iterate tracking, :271-285 -- on the device
            if sel is not None:
                sel(i % 2, i == 0, reader.select_kw()["tag"], i == 0, stream)
            else:
                be.select_best(i == 0, self.best_cost_eps, r["new_x"], r["new_u"], r["costs"], r["full_du_norm"],
                               best, flags=reader.device_flags, status=r["status"] if i == 0 else None, **reader.select_kw())
            reader.start()
            nxt = launch(i + 1) if i + 1 < self.lqr_iter else None        # overlaps the read-back
            bits, max_du_norm = reader.wait()
            any_improved = (bits & 1) != 0
            if i == 0 and not (bits & 2):
                self._c_symmetric = True
                on_symmetric()
            if self.flag_reducer is not None:          # shards agree on the batch-wide stop test (mpc.shard)
                any_improved, max_du_norm = self.flag_reducer(any_improved, max_du_norm)
            n_not_improved += 1
            if any_improved:
                n_not_improved = 0
            if self.verbose > 0:
                util.table_log("lqr", (
                    ("iter", i),
                    ("mean(cost)", best["costs"].mean().item(), "{:.4e}"),
                    ("||full_du||_max", max_du_norm, "{:.2e}"),
                    ("mean(alphas)", float(r["alphas"].mean()), "{:.2e}"),
                    ("total_qp_iters", float(r["qp_iters"].max().item())),
                ))
            if max_du_norm < self.eps or n_not_improved > self.not_improved_lim or nxt is None:
                break
            r, i = nxt, i + 1
        return best

    def _iterate_network(self, be, x_init, u, cost, dx, net, n_batch):
        """The loop when dx is an NNDynamics the kernels take (`net` = its MlpSpec) and the cost a QuadCost: per iteration
        MPC.linearize_dynamics (mpc/mpc.py:495-512), lqr_backward and lqr_forward through the network (mpc/lqr_step.py:52-261,
        module branch :223-225) as three pre-bound C calls (HipBackend.plan_network_iteration); the states of a rollout ARE
        util.get_traj of its controls through the network (:251 recomputes them), so only the first nominal needs get_traj.
        Inner iterations are never differentiated (the reference detaches them too)."""
        T = self.T
        xi = util.detach_maybe(x_init).contiguous()
        ua = util.detach_maybe(u).contiguous()
        if self.u_init is not None and ua.untyped_storage().data_ptr() == self.u_init.untyped_storage().data_ptr():
            ua = ua.clone()               # the iterations WRITE into the nominal buffers: never the caller's u_init
        xa = util.get_traj(T, ua, x_init=xi, dynamics=dx).contiguous()
        xb, ub = torch.empty_like(xa), torch.empty_like(ua)
        run, outs, vouch_c = be.plan_network_iteration(xi, cost.C, cost.c, net, self._step_options(), ((xa, ua), (xb, ub)))
        stream = torch.cuda.current_stream(xa.device).cuda_stream if xa.is_cuda else None
        r = run(0, stream)
        self._c_symmetric = False

        def on_symmetric():
            if self.lqr_iter > 2:
                vouch_c()
        return self._drive(be, lambda i: run(i % 2, stream), outs, r, xa, ua, n_batch, stream, on_symmetric)

    def _iterate_general(self, be, x_init, u, cost, dx):
        """The same loop with module-valued cost / dynamics or a slew penalty: linearisation and cost
        expansion through torch, the LQR step through `solve_lqr_subproblem`."""
        T = self.T
        best = None
        n_not_improved = 0
        self._c_symmetric = False
        for i in range(self.lqr_iter):
            u = util.detach_maybe(u)
            x = util.get_traj(T, u, x_init=x_init, dynamics=dx)
            if isinstance(dx, LinDx):
                F, f = dx.F, dx.f
            else:
                F, f = self.linearize_dynamics(x, util.detach_maybe(u), dx, diff=False)
            if isinstance(cost, QuadCost):
                C, c = cost.C, cost.c
            else:
                C, c, _ = self.approximate_cost(x, util.detach_maybe(u), cost, diff=False)
            with torch.no_grad():
                x, u, n_qp, costs, full_du_norm, mean_alphas = self.solve_lqr_subproblem(
                    x_init, C, c, F, f, cost, dx, x, u)
            n_not_improved += 1
            assert x.ndimension() == 3 and u.ndimension() == 3
            first = best is None
            if first:
                best = dict(x=torch.empty_like(x), u=torch.empty_like(u), costs=torch.empty_like(costs),
                            full_du_norm=torch.empty_like(full_du_norm))
            any_improved, max_du = be.select_best(first, self.best_cost_eps, x.contiguous(), u.contiguous(),
                                                  costs, full_du_norm, best)
            flags = torch.stack((any_improved[0].to(max_du.dtype), max_du[0])).tolist()   # the one sync
            if self.flag_reducer is not None:
                flags = list(self.flag_reducer(flags[0] != 0, flags[1]))
            if flags[0]:
                n_not_improved = 0
            max_du_norm = flags[1]
            if self.verbose > 0:
                util.table_log("lqr", (
                    ("iter", i),
                    ("mean(cost)", best["costs"].mean().item(), "{:.4e}"),
                    ("||full_du||_max", max_du_norm, "{:.2e}"),
                    ("mean(alphas)", float(mean_alphas), "{:.2e}"),
                    ("total_qp_iters", n_qp),
                ))
            if max_du_norm < self.eps or n_not_improved > self.not_improved_lim:
                break
        return best

    def _check_converged(self, full_du_norm):
        """mpc/mpc.py:321-334: raise / warn when some problem did not reach a fixed point.  Returns
        True when the unconverged problems have to be detached."""
        if not self.detach_unconverged:
            return False
        worst = float(full_du_norm.max().item())
        if worst > self.eps:
            if self.exit_unconverged:
                raise UnconvergedError(
                    "MPC: max ||du|| = %.3e > eps = %.1e after %d LQR iterations "
                    "(exit_unconverged=True)" % (worst, self.eps, self.lqr_iter))
            if self.verbose >= 0:
                print("LQR Warning: All examples did not converge to a fixed point.")
                print("Detaching and *not* backpropping through the bad examples.")
            return True
        return False

    @staticmethod
    def _wants_graph(x_init, cost, dx):
        if not torch.is_grad_enabled():
            return False
        if x_init.requires_grad or _any_requires_grad(cost) or _any_requires_grad(dx):
            return True
        # a plain callable may close over anything: keep the reference's behaviour for those
        return not all(isinstance(z, (QuadCost, LinDx, Module)) for z in (cost, dx))

    # ------------------------------------------------------------------------------------------
    def _step_options(self):
        return StepOptions(u_lower=self.u_lower, u_upper=self.u_upper, u_zero_I=self.u_zero_I,
                           delta_u=self.delta_u, linesearch_decay=self.linesearch_decay,
                           max_linesearch_iter=self.max_linesearch_iter)

    def solve_lqr_subproblem(self, x_init, C, c, F, f, cost, dynamics, x, u, no_op_forward=False):
        """Build the LQRStep for the current nominal (x,u) and apply it (mpc/mpc.py:339-361)."""
        if self.slew_rate_penalty is not None and not isinstance(cost, Module):
            return self._solve_slew_subproblem(x_init, C, c, F, f, cost, dynamics, x, u, no_op_forward)
        step = LQRStep(
            n_state=self.n_state, n_ctrl=self.n_ctrl, T=self.T,
            u_lower=self.u_lower, u_upper=self.u_upper, u_zero_I=self.u_zero_I,
            true_cost=cost, true_dynamics=dynamics, delta_u=self.delta_u,
            linesearch_decay=self.linesearch_decay, max_linesearch_iter=self.max_linesearch_iter,
            delta_space=True, current_x=x, current_u=u, back_eps=self.back_eps,
            no_op_forward=no_op_forward, c_symmetric=no_op_forward and getattr(self, "_c_symmetric", False),
            reference_du_norm=self.reference_du_norm and not no_op_forward)
        empty = torch.empty(0, dtype=x_init.dtype, device=x_init.device)
        return step(x_init, C, c, F, f if f is not None else empty)

    def _solve_slew_subproblem(self, x_init, C, c, F, f, cost, dynamics, x, u, no_op_forward):
        """slew_rate_penalty: the same LQR step on the state augmented with the previous control,
        z_t = (u_{t-1}, x_t), with 0.5*gamma*|u_t - u_{t-1}|^2 added to the stage cost
        (reference mpc/mpc.py:362-445).  Pure re-packing around the kernel; autograd reaches C, c, F, f
        through the padding ops."""
        from .dynamics import CtrlPassthroughDynamics
        T, ns, nc = self.T, self.n_state, self.n_ctrl
        B = C.size(1)
        n, na = ns + nc, ns + 2 * nc
        kw = dict(dtype=C.dtype, device=C.device)
        gI = self.slew_rate_penalty * torch.eye(nc, **kw)
        slew_C = torch.zeros(T, B, na, na, **kw)
        slew_C[:, :, :nc, :nc] = gI
        slew_C[:, :, -nc:, :nc] = -gI
        slew_C[:, :, :nc, -nc:] = -gI
        slew_C[:, :, -nc:, -nc:] = gI
        aC = slew_C + torch.nn.functional.pad(C, (nc, 0, nc, 0))
        ac = torch.cat((torch.zeros(T, B, nc, **kw), c), 2)
        carry = torch.cat((torch.zeros(nc, n, **kw), torch.eye(nc, **kw)), 1)       # u_t -> next state's u_{t-1}
        aF = torch.cat((carry.expand(T - 1, B, nc, na),
                        torch.cat((torch.zeros(T - 1, B, ns, nc, **kw), F), 3)), 2)
        af = None if f is None or f.numel() == 0 else torch.cat((torch.zeros(T - 1, B, nc, **kw), f), 2)
        if self.prev_ctrl is not None:
            prev_u = self.prev_ctrl.detach().to(**kw)
            while prev_u.ndimension() < 3:
                prev_u = prev_u.unsqueeze(0)
        else:
            prev_u = torch.zeros(1, B, nc, **kw)
        prev_u = prev_u.expand(1, B, nc)
        ax = torch.cat((torch.cat((prev_u, util.detach_maybe(u)[:-1])), x), 2)
        ax_init = torch.cat((prev_u[0], x_init), 1)
        a_dyn = None if isinstance(dynamics, LinDx) else CtrlPassthroughDynamics(dynamics)
        a_cost = QuadCost(aC, ac) if isinstance(cost, QuadCost) else SlewRateCost(cost, slew_C, ns, nc)
        step = LQRStep(
            n_state=n, n_ctrl=nc, T=T, u_lower=self.u_lower, u_upper=self.u_upper, u_zero_I=self.u_zero_I,
            true_cost=a_cost, true_dynamics=a_dyn, delta_u=self.delta_u,
            linesearch_decay=self.linesearch_decay, max_linesearch_iter=self.max_linesearch_iter,
            delta_space=True, current_x=ax, current_u=u, back_eps=self.back_eps, no_op_forward=no_op_forward,
            reference_du_norm=self.reference_du_norm and not no_op_forward)
        empty = torch.empty(0, **kw)
        out = step(ax_init, aC, ac, aF, af if af is not None else empty)
        return (out[0][:, :, nc:],) + tuple(out[1:])

    # ------------------------------------------------------------------------------------------
    def approximate_cost(self, x, u, Cf, diff=True):
        """Second-order expansion of a module cost along (x,u) via autograd (mpc/mpc.py:447-487)."""
        with torch.enable_grad():
            tau = torch.cat((x, u), dim=2).detach().requires_grad_(True)
            costs, hessians, grads = [], [], []
            for t in range(self.T):
                tau_t = tau[t]
                cost = Cf(tau_t)
                grad = torch.autograd.grad(cost.sum(), tau_t, create_graph=True, retain_graph=True)[0]
                rows = [torch.autograd.grad(grad[:, j].sum(), tau_t, retain_graph=True)[0]
                        for j in range(tau.shape[2])]
                hessian = torch.stack(rows, dim=-1)
                costs.append(cost)
                grads.append(grad - util.bmv(hessian, tau_t))
                hessians.append(hessian)
            costs, grads, hessians = torch.stack(costs), torch.stack(grads), torch.stack(hessians)
            if not diff:
                return hessians.detach(), grads.detach(), costs.detach()
            return hessians, grads, costs

    def linearize_dynamics(self, x, u, dynamics, diff):
        """F_t = [df/dx | df/du], f_t = f(x_t,u_t) - F_t [x_t;u_t] along the trajectory
        (mpc/mpc.py:490-601).  ANALYTIC uses the module's grad_input over all (T-1)*B points at
        once; AUTO_DIFF batches the reference's (T-1)*n_state backward passes into n_state."""
        T, ns, nc = self.T, self.n_state, self.n_ctrl
        B = x.shape[1]
        if self.grad_method == GradMethods.ANALYTIC:
            net = dynamics.native_net(x) if (not diff and hasattr(dynamics, "native_net") and T > 1) else None
            if net is not None and net.activation != "elu":
                # NNDynamics: forward, grad_input and the affine term in one kernel, no [N, hidden, n] intermediates
                # (elu has no grad_input in the reference, mpc/dynamics.py:113-114: left to the module to refuse)
                Fl, fl = _native.backend().mlp_linearize(net, x[:-1].reshape(-1, ns), u[:-1].reshape(-1, nc))
                return Fl.view(T - 1, B, ns, ns + nc), fl.view(T - 1, B, ns)
            # fresh leaves, as the reference (mpc/mpc.py:495-497): with diff=True the graph reaches the
            # dynamics' parameters (through new_x, R, S), not the trajectory itself.
            _x = x[:-1].reshape(-1, ns).detach().requires_grad_(True)
            _u = u[:-1].reshape(-1, nc).detach().requires_grad_(True)
            new_x = dynamics(_x, _u)
            if not diff:
                new_x, _x, _u = new_x.detach(), _x.detach(), _u.detach()
            R, S = dynamics.grad_input(_x, _u)
            f = (new_x - util.bmv(R, _x) - util.bmv(S, _u)).view(T - 1, B, ns)
            F = torch.cat((R.reshape(T - 1, B, ns, ns), S.reshape(T - 1, B, ns, nc)), 3)
            return F, f
        if self.grad_method == GradMethods.AUTO_DIFF:
            with torch.enable_grad():
                xt = x[:-1].reshape(-1, ns).detach().requires_grad_(True)
                ut = u[:-1].reshape(-1, nc).detach().requires_grad_(True)
                new_x = dynamics(xt, ut)
                Rs, Ss = [], []
                for j in range(ns):      # n_state backward passes over ALL (T-1)*B points at once
                    Rj, Sj = torch.autograd.grad(new_x[:, j].sum(), [xt, ut], retain_graph=True,
                                                 create_graph=diff)
                    Rs.append(Rj)
                    Ss.append(Sj)
                R, S = torch.stack(Rs, 1), torch.stack(Ss, 1)
                if not diff:
                    new_x, xt, ut, R, S = (z.detach() for z in (new_x, xt, ut, R, S))
                f = (new_x - util.bmv(R, xt) - util.bmv(S, ut)).view(T - 1, B, ns)
                F = torch.cat((R, S), 2).view(T - 1, B, ns, ns + nc)
            return F, f
        if self.grad_method == GradMethods.FINITE_DIFF:
            eps = 1e-4
            xt = x[:-1].reshape(-1, ns).detach()
            ut = u[:-1].reshape(-1, nc).detach()
            with torch.no_grad():
                new_x = dynamics(xt, ut)
                cols = []
                for j in range(ns + nc):
                    e = torch.zeros(ns + nc, dtype=xt.dtype, device=xt.device)
                    e[j] = eps
                    hi = dynamics(xt + e[:ns], ut + e[ns:])
                    lo = dynamics(xt - e[:ns], ut - e[ns:])
                    cols.append((hi - lo) / (2. * eps))
                Fm = torch.stack(cols, 2)
                f = (new_x - util.bmv(Fm, torch.cat((xt, ut), 1))).view(T - 1, B, ns)
            return Fm.view(T - 1, B, ns, ns + nc), f
        assert False
