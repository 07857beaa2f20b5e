#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the UNMODIFIED
reference (locuslab/mpc.pytorch, mounted read-only at /root/reference).

Run in the build container only (the GPU box has no /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Every .npz holds the inputs AND the reference's outputs, so nothing at test
time needs the reference.  Two flavours of every LQR-step output are stored:
  *_batch : the reference called once with the whole batch (its batch-global
            pnqp / line-search loops couple the problems);
  *_pp    : the reference called once per problem (n_batch = 1) -- the
            per-problem semantics the HIP kernels implement.
"""
import contextlib
import importlib.util
import io
import os
import re
import sys
import warnings

import numpy as np
import numpy.random as npr
import torch

sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/mpc"


def load_reference():
    spec = importlib.util.spec_from_file_location(
        "mpc_ref", os.path.join(REF, "__init__.py"), submodule_search_locations=[REF])
    pkg = importlib.util.module_from_spec(spec)
    sys.modules["mpc_ref"] = pkg
    spec.loader.exec_module(pkg)
    import mpc_ref.mpc as ref_mpc          # noqa
    import mpc_ref.lqr_step as ref_step    # noqa
    import mpc_ref.pnqp as ref_pnqp        # noqa
    import mpc_ref.util as ref_util        # noqa
    return ref_mpc, ref_step, ref_pnqp, ref_util


ref_mpc, ref_step, ref_pnqp, ref_util = load_reference()
QuadCost, LinDx = ref_mpc.QuadCost, ref_mpc.LinDx


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **k)
    return out, buf.getvalue()


def npy(t):
    return None if t is None else t.detach().cpu().numpy()


def save(name, **arrs):
    arrs = {k: v for k, v in arrs.items() if v is not None}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrs)
    sz = os.path.getsize(os.path.join(HERE, name + ".npz"))
    print("wrote %-36s %8.1f KB" % (name + ".npz", sz / 1024))


# --------------------------------------------------------------------------
# synthetic problem recipe
# --------------------------------------------------------------------------
def make_problem(ns, nc, T, B, dtype, seed, with_f=True, u_scale=0.3):
    g = torch.Generator().manual_seed(seed)
    n = ns + nc
    A = torch.randn(T, B, n, n, generator=g, dtype=torch.float64)
    C = A.transpose(2, 3).matmul(A)
    c = torch.randn(T, B, n, generator=g, dtype=torch.float64)
    R = torch.eye(ns, dtype=torch.float64) + 0.2 * torch.randn(T - 1, B, ns, ns, generator=g, dtype=torch.float64) / ns ** 0.5
    S = torch.randn(T - 1, B, ns, nc, generator=g, dtype=torch.float64) / ns ** 0.5
    F = torch.cat((R, S), 3)
    f = 0.1 * torch.randn(T - 1, B, ns, generator=g, dtype=torch.float64) if with_f else None
    x_init = torch.randn(B, ns, generator=g, dtype=torch.float64)
    u = u_scale * torch.randn(T, B, nc, generator=g, dtype=torch.float64)
    cast = lambda t: None if t is None else t.to(dtype).contiguous()
    return dict(C=cast(C), c=cast(c), F=cast(F), f=cast(f), x_init=cast(x_init), u=cast(u))


def slice_b(t, b, dim):
    if t is None or isinstance(t, float):
        return t
    return t.narrow(dim, b, 1).contiguous()


def run_step(p, ns, nc, T, u_lower=None, u_upper=None, u_zero_I=None, delta_u=None,
             linesearch_decay=0.2, max_linesearch_iter=10):
    """One reference LQRStep forward around the nominal (get_traj(u), u)."""
    dx = LinDx(p["F"], p["f"])
    cost = QuadCost(p["C"], p["c"])
    x = ref_util.get_traj(T, p["u"], p["x_init"], dx)
    fn = ref_step.LQRStep(
        n_state=ns, n_ctrl=nc, T=T, u_lower=u_lower, u_upper=u_upper, u_zero_I=u_zero_I,
        delta_u=delta_u, linesearch_decay=linesearch_decay, max_linesearch_iter=max_linesearch_iter,
        true_cost=cost, true_dynamics=dx, delta_space=True, current_x=x, current_u=p["u"])
    f = p["f"] if p["f"] is not None else torch.Tensor()
    (new_x, new_u, nqp, costs, fdn, mean_alpha), _ = quiet(fn, p["x_init"], p["C"], p["c"], p["F"], f)
    return dict(cur_x=x, new_x=new_x, new_u=new_u, n_qp=nqp, costs=costs, full_du_norm=fdn,
                mean_alphas=mean_alpha.reshape(1))


ONLY = None      # set from --only name1,name2: regenerate just those fixtures


def add_asymmetry(C, which, seed, scale=1.0):
    """C + scale * 2 triu(N, 1) on the problems listed in `which` (N ~ N(0,1)): a cost matrix the reference accepts and
    uses AS GIVEN -- Q = C + F'VF (mpc/lqr_step.py:68), C tau (:294), 0.5 tau'C tau (:232) -- so the solve differs from
    the one on the symmetrised matrix (the sweep's gradient C tau is then not the gradient of the cost)."""
    g = torch.Generator().manual_seed(seed)
    N = torch.randn(C.shape, generator=g, dtype=torch.float64).to(C.dtype)
    out = C.clone()
    for b in which:
        out[:, b] = C[:, b] + scale * 2.0 * torch.triu(N[:, b], 1)
    return out


def tie_case(name, which=(150, 773, 1695)):
    """The problems of the full-size box-constrained test (tests/test_gpu_fullsize.py, case "tight": bench.make_problem(12, 4,
    50, 4096, seed 0, u_scale 0.2, clamp 0.3), bounds +-0.3) on which a float32 kernel and the float64 oracle end on
    different active sets: a QP minimiser that sits on its bound to within rounding is "clamped" or "free" by the sign of a
    1e-7 gradient (mpc/pnqp.py:32), which zeroes or keeps a row of K -- a discontinuity of the reference algorithm itself.
    The fixture holds what the reference returns on exactly these problems in float32 AND in float64 (per problem), so that a
    kernel can be held to "one of the branches the reference takes"."""
    if ONLY is not None and name not in ONLY:
        return
    ns, nc, T, B = 12, 4, 50, 4096
    g = torch.Generator().manual_seed(0)
    n = ns + nc
    kw = dict(generator=g, dtype=torch.float32)
    chunks = []
    for t0 in range(0, T, 10):                       # the very draws of bench.make_problem
        A = torch.randn(min(10, T - t0), B, n, n, **kw)
        chunks.append(A.transpose(2, 3).matmul(A))
    C = torch.cat(chunks)
    c = torch.randn(T, B, n, **kw)
    R = torch.eye(ns) + 0.2 * torch.randn(T - 1, B, ns, ns, **kw) / ns ** 0.5
    S = torch.randn(T - 1, B, ns, nc, **kw) / ns ** 0.5
    F = torch.cat((R, S), 3)
    f = 0.1 * torch.randn(T - 1, B, ns, **kw)
    x_init = torch.randn(B, ns, **kw)
    u = (0.2 * torch.randn(T, B, nc, **kw)).clamp(-0.3, 0.3)
    idx = torch.tensor(list(which))
    p = dict(C=C[:, idx].contiguous(), c=c[:, idx].contiguous(), F=F[:, idx].contiguous(), f=f[:, idx].contiguous(),
             x_init=x_init[idx].contiguous(), u=u[:, idx].contiguous())
    outs32, outs64 = [], []
    to64 = lambda t: t.double() if torch.is_tensor(t) else t
    for b in range(len(which)):
        pb = {k: slice_b(v, b, 0 if k == "x_init" else 1) for k, v in p.items()}
        outs32.append(run_step(pb, ns, nc, T, u_lower=-0.3, u_upper=0.3))
        outs64.append(run_step({k: to64(v) for k, v in pb.items()}, ns, nc, T, u_lower=-0.3, u_upper=0.3))
    cat = lambda outs, key, dim: npy(torch.cat([o[key] for o in outs], dim))
    save(name, meta=np.array([ns, nc, T, len(which), -1, 10], dtype=np.int64), decay=np.array([0.2]), delta_u=np.array([np.nan]),
         which=np.array(which), C=npy(p["C"]), c=npy(p["c"]), F=npy(p["F"]), f=npy(p["f"]), x_init=npy(p["x_init"]),
         cur_u=npy(p["u"]), cur_x=cat(outs32, "cur_x", 1), u_lower=np.array([-0.3]), u_upper=np.array([0.3]),
         new_x_pp=cat(outs32, "new_x", 1), new_u_pp=cat(outs32, "new_u", 1), costs_pp=cat(outs32, "costs", 0),
         alphas_pp=cat(outs32, "mean_alphas", 0), n_qp_pp=cat(outs32, "n_qp", 0),
         new_x_ref64=cat(outs64, "new_x", 1), new_u_ref64=cat(outs64, "new_u", 1), costs_ref64=cat(outs64, "costs", 0),
         alphas_ref64=cat(outs64, "mean_alphas", 0))


def step_case(name, ns, nc, T, B, dtype, seed, with_f=True, bounds=None, mask_seed=None,
              delta_u=None, decay=0.2, max_ls=10, u_scale=0.3, indef=0.0, asym=None, asym_scale=1.0, zero_ctrl=None):
    if ONLY is not None and name not in ONLY:
        return
    p = make_problem(ns, nc, T, B, dtype, seed, with_f, u_scale)
    if asym is not None:
        p["C"] = add_asymmetry(p["C"], asym, seed + 500, asym_scale)
    if zero_ctrl is not None:
        # control `a` of the listed problems enters neither the cost nor the dynamics: Quu is singular at every timestep
        # and the reference's pinverse (mpc/lqr_step.py:88-94) returns a zero gain for it
        for b, a in zero_ctrl:
            p["C"][:, b, ns + a, :] = 0
            p["C"][:, b, :, ns + a] = 0
            p["F"][:, b, :, ns + a] = 0
    if indef:
        # a NON-convex stage cost in the states: the Newton step is no longer a descent step for
        # every problem, so the line search of mpc/lqr_step.py:176-252 really backtracks
        p["C"][:, :, :ns, :ns] -= indef * torch.eye(ns, dtype=p["C"].dtype)
    g = torch.Generator().manual_seed(seed + 1000)
    u_lower = u_upper = None
    if bounds == "tensor":
        u_lower = -torch.rand(T, B, nc, generator=g, dtype=torch.float64).to(dtype)
        u_upper = torch.rand(T, B, nc, generator=g, dtype=torch.float64).to(dtype)
        # nominal controls must be feasible for the delta-space QP bounds to make sense
        p["u"] = torch.max(torch.min(p["u"], u_upper), u_lower)
    elif isinstance(bounds, float):
        u_lower, u_upper = -bounds, bounds
        p["u"] = p["u"].clamp(-bounds, bounds)
    mask = None
    if mask_seed is not None:
        mask = torch.rand(T, B, nc, generator=torch.Generator().manual_seed(mask_seed)) < 0.35
        p["u"] = p["u"] * (~mask).to(dtype)
    kw = dict(u_lower=u_lower, u_upper=u_upper, u_zero_I=mask, delta_u=delta_u,
              linesearch_decay=decay, max_linesearch_iter=max_ls)
    out_b = run_step(p, ns, nc, T, **kw)
    outs = []
    for b in range(B):
        pb = {k: slice_b(v, b, 0 if k == "x_init" else 1) for k, v in p.items()}
        kwb = dict(kw)
        kwb["u_lower"] = slice_b(u_lower, b, 1)
        kwb["u_upper"] = slice_b(u_upper, b, 1)
        kwb["u_zero_I"] = slice_b(mask, b, 1)
        try:
            outs.append(run_step(pb, ns, nc, T, **kwb))
        except RuntimeError:
            # torch>=2 rejects the reference's uint8 mask in util.bdiag when the masked
            # assignment degenerates to a 1-element fill (nc=1, n_batch=1).  Run the problem
            # twice side by side instead: two identical elements walk the batch-global loops
            # exactly like one.
            dup = lambda t, d: None if (t is None or isinstance(t, float)) else torch.cat((t, t), d)
            pb2 = {k: dup(v, 0 if k == "x_init" else 1) for k, v in pb.items()}
            kw2 = dict(kwb)
            for k in ("u_lower", "u_upper", "u_zero_I"):
                kw2[k] = dup(kwb[k], 1)
            o2 = run_step(pb2, ns, nc, T, **kw2)
            outs.append({k: (v if k in ("n_qp", "mean_alphas") else v.narrow(0 if v.dim() == 1 else 1, 0, 1))
                         for k, v in o2.items()})
    cat = lambda key, dim: torch.cat([o[key] for o in outs], dim)
    # For f32 cases also store the reference run in DOUBLE precision on the same (f32-valued)
    # inputs, per problem: the "reference truth" the stated fp32 tolerance is measured against.
    ref64 = {}
    if dtype == torch.float32:
        outs64 = []
        to64 = lambda t: t.double() if torch.is_tensor(t) and t.is_floating_point() else t
        for b in range(B):
            pb = {k: to64(slice_b(v, b, 0 if k == "x_init" else 1)) for k, v in p.items()}
            kwb = dict(kw)
            kwb["u_lower"] = to64(slice_b(u_lower, b, 1))
            kwb["u_upper"] = to64(slice_b(u_upper, b, 1))
            kwb["u_zero_I"] = slice_b(mask, b, 1)
            outs64.append(run_step(pb, ns, nc, T, **kwb))
        c64 = lambda key, dim: npy(torch.cat([o[key] for o in outs64], dim))
        ref64 = dict(new_x_ref64=c64("new_x", 1), new_u_ref64=c64("new_u", 1), costs_ref64=c64("costs", 0),
                     full_du_norm_ref64=c64("full_du_norm", 0))
    meta = np.array([ns, nc, T, B, -1 if delta_u is None else 1, max_ls], dtype=np.int64)
    save(name,
         meta=meta, decay=np.array([decay]), delta_u=np.array([np.nan if delta_u is None else delta_u]),
         C=npy(p["C"]), c=npy(p["c"]), F=npy(p["F"]), f=npy(p["f"]), x_init=npy(p["x_init"]),
         cur_u=npy(p["u"]), cur_x=npy(out_b["cur_x"]),
         u_lower=(np.array([u_lower]) if isinstance(u_lower, float) else npy(u_lower)),
         u_upper=(np.array([u_upper]) if isinstance(u_upper, float) else npy(u_upper)),
         u_zero_I=None if mask is None else npy(mask).astype(np.uint8),
         new_x_batch=npy(out_b["new_x"]), new_u_batch=npy(out_b["new_u"]), costs_batch=npy(out_b["costs"]),
         full_du_norm_batch=npy(out_b["full_du_norm"]), mean_alphas_batch=npy(out_b["mean_alphas"]),
         n_qp_batch=npy(out_b["n_qp"]),
         new_x_pp=npy(cat("new_x", 1)), new_u_pp=npy(cat("new_u", 1)), costs_pp=npy(cat("costs", 0)),
         full_du_norm_pp=npy(cat("full_du_norm", 0)), alphas_pp=npy(cat("mean_alphas", 0)),
         n_qp_pp=npy(cat("n_qp", 0)), **ref64)


# --------------------------------------------------------------------------
# full MPC.forward solves (reference tests + the notebook known-answer table)
# --------------------------------------------------------------------------
def parse_table(text):
    rows = []
    for line in text.splitlines():
        m = re.match(r"\|\s*(\d+)\s*\|\s*([^|]+)\|\s*([^|]+)\|\s*([^|]+)\|\s*([^|]+)\|", line)
        if m:
            num = lambda s_: float(re.search(r"[-+0-9.eE]+", s_.replace("tensor", "")).group(0))
            rows.append([num(m.group(i)) for i in range(1, 6)])
    init = re.search(r"Initial mean\(cost\): ([0-9.eE+-]+)", text)
    return np.array(rows), np.array([float(init.group(1))] if init else [])


def mpc_case(name, ns, nc, T, C, c, F, f, x_init, u_lower, u_upper, **kw):
    ctrl = ref_mpc.MPC(ns, nc, T, u_lower=u_lower, u_upper=u_upper, verbose=1, **kw)
    (x, u, costs), txt = quiet(ctrl, x_init, QuadCost(C, c), LinDx(F, f))
    table, init = parse_table(txt)
    kwn = {("kw_" + k): np.array([v if v is not None else np.nan], dtype=np.float64) for k, v in kw.items()
           if isinstance(v, (int, float, bool)) or v is None}
    save(name, meta=np.array([ns, nc, T, C.shape[1]]), C=npy(C), c=npy(c), F=npy(F), f=npy(f), x_init=npy(x_init),
         u_lower=None if u_lower is None else (np.array([u_lower]) if isinstance(u_lower, float) else npy(u_lower)),
         u_upper=None if u_upper is None else (np.array([u_upper]) if isinstance(u_upper, float) else npy(u_upper)),
         x=npy(x), u=npy(u), costs=npy(costs), table=table, init_cost=init, **kwn)


def env_case(name, kind, T, B, lqr_iter, seed):
    """BASELINE.json configs 2 / 3: iLQR on the reference's own simulator dynamics (AUTO_DIFF
    linearisation, module rollout in the line search), double precision, a few problems."""
    import importlib
    sys.modules.setdefault("mpc", sys.modules["mpc_ref"])        # env_dx does `from mpc import util`
    sys.modules.setdefault("mpc.util", ref_util)
    mod = importlib.import_module("mpc_ref.env_dx." + kind)
    dx = getattr(mod, "PendulumDx" if kind == "pendulum" else "CartpoleDx")()
    dx.params = dx.params.double()
    ns, nc = dx.n_state, dx.n_ctrl
    g = torch.Generator().manual_seed(seed)
    if kind == "pendulum":
        th = (torch.rand(B, generator=g, dtype=torch.float64) - 0.5) * np.pi
        thd = (torch.rand(B, generator=g, dtype=torch.float64) - 0.5) * 2.0
        x_init = torch.stack((torch.cos(th), torch.sin(th), thd), 1)
    else:
        th = (torch.rand(B, generator=g, dtype=torch.float64) - 0.5) * 0.6
        z = 0.2 * torch.randn(B, 3, generator=g, dtype=torch.float64)
        x_init = torch.stack((z[:, 0], z[:, 1], torch.cos(th), torch.sin(th), z[:, 2]), 1)
    q, p_ = dx.get_true_obj()
    q, p_ = q.double(), p_.double()
    Q = torch.diag(q).unsqueeze(0).unsqueeze(0).repeat(T, B, 1, 1)
    pp = p_.unsqueeze(0).repeat(T, B, 1)
    ctrl = ref_mpc.MPC(ns, nc, T, u_lower=dx.lower, u_upper=dx.upper, lqr_iter=lqr_iter, verbose=-1,
                       exit_unconverged=False, detach_unconverged=False, linesearch_decay=dx.linesearch_decay,
                       max_linesearch_iter=dx.max_linesearch_iter, grad_method=ref_mpc.GradMethods.AUTO_DIFF,
                       eps=dx.mpc_eps)
    (x, u, costs), _ = quiet(ctrl, x_init, QuadCost(Q, pp), dx)
    save(name, meta=np.array([ns, nc, T, B, lqr_iter]), x_init=npy(x_init), Q=npy(Q), p=npy(pp),
         lower=np.array([dx.lower]), upper=np.array([dx.upper]), decay=np.array([dx.linesearch_decay]),
         max_ls=np.array([dx.max_linesearch_iter]), eps=np.array([dx.mpc_eps]),
         x=npy(x), u=npy(u), costs=npy(costs))


def _ref_env(kind, simple=True, params=None):
    import importlib
    sys.modules.setdefault("mpc", sys.modules["mpc_ref"])
    sys.modules.setdefault("mpc.util", ref_util)
    if kind == "pendulum":
        dx = importlib.import_module("mpc_ref.env_dx.pendulum").PendulumDx(params=params, simple=simple)
    else:
        dx = importlib.import_module("mpc_ref.env_dx.cartpole").CartpoleDx(params=params)
    dx.params = dx.params.double()
    return dx


def env_lin_case(name, kind, T, B, seed, simple=True, params=None):
    """The reference's own simulator modules at random points: next state, and F, f of
    MPC.linearize_dynamics (GradMethods.AUTO_DIFF, mpc/mpc.py:514-549).  Controls reach past the
    modules' clamp on purpose.  Also ONE reference LQRStep with the module as true_dynamics."""
    dx = _ref_env(kind, simple, None if params is None else torch.tensor(params))
    ns, nc = dx.n_state, dx.n_ctrl
    g = torch.Generator().manual_seed(seed)
    f64 = torch.float64
    umax = dx.upper
    u = (torch.rand(T, B, nc, generator=g, dtype=f64) - 0.5) * 2.6 * umax
    th = (torch.rand(B, generator=g, dtype=f64) - 0.5) * 2 * np.pi
    if kind == "pendulum":
        x_init = torch.stack((torch.cos(th), torch.sin(th), torch.randn(B, generator=g, dtype=f64)), 1)
    else:
        z = torch.randn(B, 3, generator=g, dtype=f64)
        x_init = torch.stack((z[:, 0], z[:, 1], torch.cos(th), torch.sin(th), z[:, 2]), 1)
        u = u * 0.05
    x = ref_util.get_traj(T, u, x_init, dx)
    # off-manifold states as well (cos/sin not normalised): scale a few
    x = x * (1.0 + 0.1 * torch.randn(T, B, 1, generator=g, dtype=f64))
    ctrl = ref_mpc.MPC(ns, nc, T, u_lower=dx.lower, u_upper=dx.upper, lqr_iter=1, verbose=-1,
                       grad_method=ref_mpc.GradMethods.AUTO_DIFF)
    F, f = ctrl.linearize_dynamics(x, u, dx, diff=False)
    nxt = dx(x[:-1].reshape(-1, ns), u[:-1].reshape(-1, nc)).view(T - 1, B, ns)
    # one LQR step around a proper nominal, true_dynamics = the module
    u0 = 0.3 * (torch.rand(T, B, nc, generator=g, dtype=f64) - 0.5) * umax * (0.05 if kind == "cartpole" else 1.0)
    x0 = ref_util.get_traj(T, u0, x_init, dx)
    F0, f0 = ctrl.linearize_dynamics(x0, u0, dx, diff=False)
    q, p_ = dx.get_true_obj()
    Q = torch.diag(q.double()).unsqueeze(0).unsqueeze(0).repeat(T, B, 1, 1)
    pp = p_.double().unsqueeze(0).repeat(T, B, 1)
    fn = ref_step.LQRStep(n_state=ns, n_ctrl=nc, T=T, u_lower=dx.lower, u_upper=dx.upper,
                          linesearch_decay=dx.linesearch_decay, max_linesearch_iter=dx.max_linesearch_iter,
                          true_cost=QuadCost(Q, pp), true_dynamics=dx, delta_space=True, current_x=x0,
                          current_u=u0)
    (new_x, new_u, nqp, costs, fdn, mean_alpha), _ = quiet(fn, x_init, Q, pp, F0, f0)
    save(name, meta=np.array([ns, nc, T, B]), params=npy(dx.params), simple=np.array([int(simple)]),
         x=npy(x), u=npy(u), next=npy(nxt), F=npy(F), f=npy(f), x_init=npy(x_init),
         lower=np.array([dx.lower]), upper=np.array([dx.upper]), decay=np.array([dx.linesearch_decay]),
         max_ls=np.array([dx.max_linesearch_iter]), Q=npy(Q), p=npy(pp),
         step_cur_x=npy(x0), step_cur_u=npy(u0), step_F=npy(F0), step_f=npy(f0), step_new_x=npy(new_x),
         step_new_u=npy(new_u), step_costs=npy(costs), step_full_du_norm=npy(fdn))


def nn_case(name, seed, ns, nc, hidden, act, passthrough, T, B, bound, decay=0.2, max_ls=10, lqr_iter=6,
            w_scale=1.0):
    """mpc.dynamics.NNDynamics (mpc/dynamics.py:15-128) as the dynamics: the module at random points (forward and
    grad_input), MPC.linearize_dynamics(ANALYTIC) along a nominal, ONE reference LQRStep with the module as
    true_dynamics (mpc/lqr_step.py:223-225), and a whole MPC.forward solve."""
    from mpc_ref.dynamics import NNDynamics
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed)
    f64 = torch.float64
    dyn = NNDynamics(ns, nc, list(hidden), activation=act, passthrough=passthrough).double()
    with torch.no_grad():
        for fc in dyn.fcs:
            fc.weight.mul_(w_scale)
    n = ns + nc
    N = 24
    px = torch.randn(N, ns, generator=g, dtype=f64)
    pu = torch.randn(N, nc, generator=g, dtype=f64)
    with torch.no_grad():
        pnext = dyn(px, pu)
        pR, pS = dyn.grad_input(px, pu)
    x_init = torch.randn(B, ns, generator=g, dtype=f64)
    u0 = 0.3 * torch.randn(T, B, nc, generator=g, dtype=f64)
    A = torch.randn(T, B, n, n, generator=g, dtype=f64)
    C = A.transpose(2, 3).matmul(A) + 0.1 * torch.eye(n, dtype=f64)
    c = torch.randn(T, B, n, generator=g, dtype=f64)
    with torch.no_grad():
        x0 = ref_util.get_traj(T, u0, x_init, dyn)
    lo, hi = (None, None) if bound is None else (-float(bound), float(bound))
    ctrl = ref_mpc.MPC(ns, nc, T, u_lower=lo, u_upper=hi, lqr_iter=lqr_iter, verbose=-1,
                       exit_unconverged=False, detach_unconverged=False, linesearch_decay=decay,
                       max_linesearch_iter=max_ls, grad_method=ref_mpc.GradMethods.ANALYTIC, u_init=u0.clone())
    F0, f0 = ctrl.linearize_dynamics(x0, u0, dyn, diff=False)
    fn = ref_step.LQRStep(n_state=ns, n_ctrl=nc, T=T, u_lower=lo, u_upper=hi, linesearch_decay=decay,
                          max_linesearch_iter=max_ls, true_cost=QuadCost(C, c), true_dynamics=dyn,
                          delta_space=True, current_x=x0, current_u=u0)
    with torch.no_grad():
        (new_x, new_u, nqp, costs, fdn, mean_alpha), _ = quiet(fn, x_init, C, c, F0, f0)
    (sx, su, scosts), _ = quiet(ctrl, x_init, QuadCost(C, c), dyn)
    arrs = {}
    for i, fc in enumerate(dyn.fcs):
        arrs["W%d" % i], arrs["b%d" % i] = npy(fc.weight), npy(fc.bias)
    save(name, meta=np.array([ns, nc, T, B, lqr_iter]),
         nn_meta=np.array([len(dyn.fcs), ("sigmoid", "relu", "elu").index(act), int(passthrough)]),
         bound=np.array([np.nan if bound is None else bound]), decay=np.array([decay]), max_ls=np.array([max_ls]),
         px=npy(px), pu=npy(pu), pnext=npy(pnext), pR=npy(pR), pS=npy(pS),
         x_init=npy(x_init), C=npy(C), c=npy(c), step_cur_x=npy(x0), step_cur_u=npy(u0), step_F=npy(F0),
         step_f=npy(f0), step_new_x=npy(new_x), step_new_u=npy(new_u), step_costs=npy(costs),
         step_full_du_norm=npy(fdn), solve_x=npy(sx), solve_u=npy(su), solve_costs=npy(scosts), **arrs)


def slew_case(name, seed, ns=2, nc=2, T=4, B=2, gamma=1.0, prev=False):
    """tests/test_mpc.py:652-744 (test_lqr_backward_cost_nn_dynamics_module_constrained_slew): MPC with
    slew_rate_penalty on an NNDynamics module, box constraints, ANALYTIC linearisation; the solve and
    the gradients of a fixed linear functional of (x,u) w.r.t. C, c, x_init and the first bias."""
    from mpc_ref.dynamics import NNDynamics
    npr.seed(seed)
    torch.manual_seed(seed)
    n = ns + nc
    C = 10. * npr.randn(T, B, n, n)
    C = np.matmul(C.transpose(0, 1, 3, 2), C)
    c = 10. * npr.randn(T, B, n)
    x_init = npr.randn(B, ns)
    lo, hi = -np.ones((T, B, nc)), np.ones((T, B, nc))
    dyn = NNDynamics(ns, nc, [10, 10], activation='sigmoid').double()
    prev_ctrl = torch.tensor(0.3 * npr.randn(B, nc)) if prev else None
    wx, wu = npr.randn(T, B, ns), npr.randn(T, B, nc)
    tC, tc, tx0 = (torch.tensor(a, requires_grad=True) for a in (C, c, x_init))
    ctrl = ref_mpc.MPC(ns, nc, T, torch.tensor(lo), torch.tensor(hi), None, lqr_iter=40, verbose=-1,
                       max_linesearch_iter=1, grad_method=ref_mpc.GradMethods.ANALYTIC,
                       slew_rate_penalty=gamma, prev_ctrl=prev_ctrl, exit_unconverged=False)
    (x, u, costs), _ = quiet(ctrl, tx0, QuadCost(tC, tc), dyn)
    loss = (x * torch.tensor(wx)).sum() + (u * torch.tensor(wu)).sum()
    gC, gc, gx0, gb0 = torch.autograd.grad(loss, [tC, tc, tx0, dyn.fcs[0].bias])
    arrs = {}
    for i, fc in enumerate(dyn.fcs):
        arrs["W%d" % i], arrs["b%d" % i] = npy(fc.weight), npy(fc.bias)
    save(name, meta=np.array([ns, nc, T, B]), gamma=np.array([gamma]), C=C, c=c, x_init=x_init, lo=lo, hi=hi,
         prev_ctrl=npy(prev_ctrl), wx=wx, wu=wu, x=npy(x), u=npy(u), costs=npy(costs), gC=npy(gC), gc=npy(gc),
         gx0=npy(gx0), gb0=npy(gb0), **arrs)


def module_cost_case(name, seed, ns=3, nc=2, T=5, B=3, bound=0.6, lqr_iter=25):
    """A non-quadratic nn.Module cost through the reference's MPC.forward: every iteration expands it with
    approximate_cost (mpc/mpc.py:447-487, called at :261), the line search prices trials with the module itself
    (mpc/lqr_step.py:233-234) and the final differentiable expansion (:316) carries the gradient of a fixed linear
    functional of (x, u) into the module's parameters.  LinDx dynamics, box constraints, float64."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
    import envs
    npr.seed(seed)
    torch.manual_seed(seed)
    n = ns + nc
    R = np.tile(np.eye(ns) + 0.2 * npr.randn(ns, ns), (T - 1, B, 1, 1))
    S = np.tile(npr.randn(ns, nc), (T - 1, B, 1, 1))
    F = torch.tensor(np.concatenate((R, S), axis=3))
    f = torch.tensor(0.1 * npr.randn(T - 1, B, ns))
    x_init = torch.tensor(npr.randn(B, ns))
    wx, wu = npr.randn(T, B, ns), npr.randn(T, B, nc)
    cost = envs.SmoothCost(n, seed=seed)
    ctrl = ref_mpc.MPC(ns, nc, T, u_lower=-bound, u_upper=bound, lqr_iter=lqr_iter, verbose=-1, n_batch=B,
                       exit_unconverged=False, detach_unconverged=False, eps=1e-9)
    (x, u, costs), _ = quiet(ctrl, x_init, cost, LinDx(F, f))
    loss = (x * torch.tensor(wx)).sum() + (u * torch.tensor(wu)).sum()
    g_goal, g_P = torch.autograd.grad(loss, [cost.goal, cost.P])
    # one expansion on its own, at the solution: (C, c, stage costs) of approximate_cost
    Cq, cq, sc = ctrl.approximate_cost(x.detach(), u.detach(), cost, diff=False)
    save(name, meta=np.array([ns, nc, T, B]), seed=np.array([seed]), bound=np.array([bound]), lqr_iter=np.array([lqr_iter]),
         F=npy(F), f=npy(f), x_init=npy(x_init), wx=wx, wu=wu, x=npy(x), u=npy(u), costs=npy(costs),
         g_goal=npy(g_goal), g_P=npy(g_P), approx_C=npy(Cq), approx_c=npy(cq), approx_costs=npy(sc))


def gen_mpc_cases():
    # (1) notebook known answer: examples/Time Varying Linear-Quadratic Control.ipynb cell 1
    torch.manual_seed(0)
    B, ns, nc, T = 2, 3, 4, 5
    n = ns + nc
    C = torch.randn(T * B, n, n)
    C = torch.bmm(C, C.transpose(1, 2)).view(T, B, n, n)
    c = torch.randn(T, B, n)
    R = (torch.eye(ns) + 0.2 * torch.randn(ns, ns)).repeat(T, B, 1, 1)
    S = torch.randn(T, B, ns, nc)
    F = torch.cat((R, S), dim=3)
    x_init = torch.randn(B, ns)
    u_lower = -torch.rand(T, B, nc)
    u_upper = torch.rand(T, B, nc)
    mpc_case("mpc_notebook_tvlq", ns, nc, T, C, c, F, None, x_init, u_lower, u_upper,
             lqr_iter=20, backprop=False, exit_unconverged=False)

    # (2,3) tests/test_mpc.py:91-149 test_lqr_linear_unbounded (inputs :92-111)
    def npr_problem(seed, B, ns, nc, T, S_scale=1.0):
        npr.seed(seed)
        n = ns + nc
        C = npr.randn(T, B, n, n)
        C = np.matmul(C.transpose(0, 1, 3, 2), C)
        c = npr.randn(T, B, n)
        R = np.tile(np.eye(ns) + 0.2 * np.random.randn(ns, ns), (T, B, 1, 1))
        S = S_scale * np.tile(np.random.randn(ns, nc), (T, B, 1, 1))
        F = np.concatenate((R, S), axis=3)
        f = np.tile(npr.randn(ns), (T, B, 1))
        x_init = npr.randn(B, ns)
        return [torch.tensor(a).double() for a in (C, c, F, f, x_init)]

    C, c, F, f, x_init = npr_problem(1, 2, 3, 4, 5)
    big = 1e4 * torch.ones(5, 2, 4).double()
    mpc_case("mpc_linear_unbounded_big_bounds", 3, 4, 5, C, c, F, f, x_init, -big, big,
             lqr_iter=10, backprop=False, exit_unconverged=True)
    mpc_case("mpc_linear_unbounded_none", 3, 4, 5, C, c, F, f, x_init, None, None,
             lqr_iter=10, backprop=False, exit_unconverged=False)

    # (4) tests/test_mpc.py:152-194 test_lqr_linear_bounded
    C, c, F, f, x_init = npr_problem(1, 2, 3, 4, 5)
    u_lower = torch.tensor(-npr.random((5, 2, 4))).double()
    u_upper = torch.tensor(npr.random((5, 2, 4))).double()
    mpc_case("mpc_linear_bounded", 3, 4, 5, C, c, F, f, x_init, u_lower, u_upper,
             lqr_iter=20, backprop=False, exit_unconverged=False)

    # (5) tests/test_mpc.py:197-240 test_lqr_linear_bounded_delta
    C, c, F, f, x_init = npr_problem(1, 2, 3, 4, 5, S_scale=0.01)
    u_lower = torch.tensor(-npr.random((5, 2, 4))).double()
    u_upper = torch.tensor(npr.random((5, 2, 4))).double()
    mpc_case("mpc_linear_bounded_delta", 3, 4, 5, C, c, F, f, x_init, u_lower, u_upper,
             lqr_iter=1, delta_u=0.1, backprop=False, exit_unconverged=False)

    # (6) tests/test_mpc.py:243-299 test_lqr_cuda_singleton (nc = 1 scalar branches), on CPU
    C, c, F, f, x_init = npr_problem(1, 5, 3, 1, 5)
    big = 1e4 * torch.ones(5, 5, 1).double()
    mpc_case("mpc_singleton_big_bounds", 3, 1, 5, C, c, F, f, x_init, -big, big,
             lqr_iter=10, backprop=False, exit_unconverged=False)
    mpc_case("mpc_singleton_none", 3, 1, 5, C, c, F, f, x_init, None, None,
             lqr_iter=10, backprop=False, exit_unconverged=False)


# --------------------------------------------------------------------------
# KKT backward (LQRStepFn.backward) goldens
# --------------------------------------------------------------------------
def grad_case(name, ns, nc, T, B, dtype, seed, beta, with_f=True, lqr_iter=30, scale=1.0, asym=None):
    """Solve to the fixed point with the reference, then push random (dl_dx, dl_du) through
    LQRStepFn.backward.  Stores the fixed point and all five gradients."""
    if ONLY is not None and name not in ONLY:
        return
    p = make_problem(ns, nc, T, B, dtype, seed, with_f)
    if asym is not None:
        p["C"] = add_asymmetry(p["C"], asym, seed + 500, 0.3)
    C = (scale * p["C"]).requires_grad_(True)
    c = (scale * p["c"]).requires_grad_(True)
    F = p["F"].clone().requires_grad_(True)
    f = p["f"].clone().requires_grad_(True) if with_f else None
    x_init = p["x_init"].clone().requires_grad_(True)
    bounds = dict(u_lower=None, u_upper=None) if beta is None else dict(u_lower=-beta, u_upper=beta)
    ctrl = ref_mpc.MPC(ns, nc, T, lqr_iter=lqr_iter, verbose=-1, exit_unconverged=False,
                       detach_unconverged=False, eps=1e-9 if dtype == torch.float64 else 1e-4,
                       **bounds)
    (x, u, costs), _ = quiet(ctrl, x_init, QuadCost(C, c), LinDx(F, f))
    g = torch.Generator().manual_seed(seed + 77)
    gx = torch.randn(x.shape, generator=g, dtype=torch.float64).to(dtype)
    gu = torch.randn(u.shape, generator=g, dtype=torch.float64).to(dtype)
    loss = (x * gx).sum() + (u * gu).sum()
    ins = [x_init, C, c, F] + ([f] if with_f else [])
    grads = torch.autograd.grad(loss, ins)
    save(name, meta=np.array([ns, nc, T, B]), beta=np.array([np.nan if beta is None else beta]),
         C=npy(C), c=npy(c), F=npy(F), f=npy(f), x_init=npy(x_init),
         x=npy(x), u=npy(u), dl_dx=npy(gx), dl_du=npy(gu),
         dx_init=npy(grads[0]), dC=npy(grads[1]), dc=npy(grads[2]), dF=npy(grads[3]),
         df=npy(grads[4]) if with_f else None)


def du_norm_case(name, ns=3, nc=2, T=6, B=8, beta=0.4, lqr_iter=3, eps=2e-2, seeds=range(200, 400)):
    """The reference's `full_du_norm` at n_batch > 1 (mpc/lqr_step.py:243-245: a transpose in front of the reshape mixes the
    problems of a batch) and what hangs on it in MPC.forward: the eps exit (mpc/mpc.py:299) and the detach mask (:321-334).  A
    short solve (lqr_iter = 3) with a loose eps, so that some ROWS of the mixed-up vector are below eps and others are not -- and
    the mask differs from the one each problem's own norm would give.  Stored: the vector of every iteration, the solve, the
    mask, and the gradients of a random linear loss (zero where the mask detaches)."""
    if ONLY is not None and name not in ONLY:
        return
    for seed in seeds:
        p = make_problem(ns, nc, T, B, torch.float64, seed, True, u_scale=0.05)
        # half of the problems start from (nearly) their fixed point: solve them first, restart there
        warm = ref_mpc.MPC(ns, nc, T, u_lower=-beta, u_upper=beta, lqr_iter=40, verbose=-1, exit_unconverged=False,
                           detach_unconverged=False, eps=1e-10, n_batch=B, u_init=p["u"])
        (_, u_star, _), _ = quiet(warm, p["x_init"], QuadCost(p["C"], p["c"]), LinDx(p["F"], p["f"]))
        u0 = p["u"].clone()
        u0[:, ::2] = u_star[:, ::2].detach()
        C = p["C"].clone().requires_grad_(True)
        c = p["c"].clone().requires_grad_(True)
        x_init = p["x_init"].clone().requires_grad_(True)
        ctrl = ref_mpc.MPC(ns, nc, T, u_lower=-beta, u_upper=beta, lqr_iter=lqr_iter, verbose=-1, exit_unconverged=False,
                           detach_unconverged=True, eps=eps, n_batch=B, u_init=u0)
        seen = []
        inner = ctrl.solve_lqr_subproblem

        def spy(*a, **k):
            out = inner(*a, **k)
            if len(out) == 6:
                seen.append((out[4].detach().clone(), out[3].detach().clone()))
            return out
        ctrl.solve_lqr_subproblem = spy
        (x, u, costs), _ = quiet(ctrl, x_init, QuadCost(C, c), LinDx(p["F"], p["f"]))
        # the best-iterate bookkeeping of mpc/mpc.py:271-285 on the recorded vectors
        best_n, best_c = seen[0][0].clone(), seen[0][1].clone()
        for fdn, cs in seen[1:]:
            take = cs <= best_c + ctrl.best_cost_eps
            best_n[take], best_c[take] = fdn[take], cs[take]
        keep = best_n < eps
        # ... against each problem's own norm of the same iterations (what n_batch = 1 calls would have seen)
        g = torch.Generator().manual_seed(seed + 77)
        gx = torch.randn(x.shape, generator=g, dtype=torch.float64)
        gu = torch.randn(u.shape, generator=g, dtype=torch.float64)
        grads = torch.autograd.grad((x * gx).sum() + (u * gu).sum(), [x_init, C, c], allow_unused=True)
        dead = np.array([float(grads[2][:, b].abs().max()) == 0.0 for b in range(B)])
        if not (keep.any() and (~keep).any()):
            continue
        assert (dead == ~keep.numpy()).all(), (dead, keep)
        save(name, meta=np.array([ns, nc, T, B, lqr_iter, seed]), beta=np.array([beta]), eps=np.array([eps]),
             C=npy(C), c=npy(c), F=npy(p["F"]), f=npy(p["f"]), x_init=npy(x_init), u_init=npy(u0),
             x=npy(x), u=npy(u), costs=npy(costs), full_du_norm_iters=np.stack([npy(a) for a, _ in seen]),
             costs_iters=np.stack([npy(b_) for _, b_ in seen]), keep=keep.numpy(), dl_dx=npy(gx), dl_du=npy(gu),
             dx_init=npy(grads[0]), dC=npy(grads[1]), dc=npy(grads[2]))
        print("   %s: seed %d, %d iterations, keep mask %s" % (name, seed, len(seen), keep.numpy().astype(int).tolist()))
        return
    raise RuntimeError("du_norm_case: no seed gave a mixed mask")


def jacobian_case(name, beta):
    """tests/test_mpc.py:303-395 / :398-500 inputs; full du/d{C,c,F,f,x_init} Jacobians through
    the reference's autograd (the numdifftools oracle is not installed here)."""
    npr.seed(0)
    torch.manual_seed(0)
    B, ns, nc, T = 1, 2, 2, 3
    n = ns + nc
    C = 10. * npr.randn(T, B, n, n)
    C = np.matmul(C.transpose(0, 1, 3, 2), C)
    c = 10. * npr.randn(T, B, n)
    x_init = npr.randn(B, ns)
    F = npr.randn(T - 1, B, ns, n)
    f = npr.randn(T - 1, B, ns)
    tens = [torch.tensor(a).double().requires_grad_(True) for a in (C, c, x_init, F, f)]
    _C, _c, _x, _F, _f = tens
    lo = -beta * torch.ones(T, B, nc).double()
    hi = beta * torch.ones(T, B, nc).double()
    ctrl = ref_mpc.MPC(ns, nc, T, lo, hi, None, lqr_iter=20, verbose=-1, exit_unconverged=False)
    (x, u, _), _ = quiet(ctrl, _x, QuadCost(_C, _c), LinDx(_F, _f))
    uf = u.view(-1)
    J = {k: [] for k in ("dC", "dc", "dx_init", "dF", "df")}
    for i in range(len(uf)):
        gs = torch.autograd.grad(uf[i], tens, retain_graph=True)
        for k, gi in zip(("dC", "dc", "dx_init", "dF", "df"), gs):
            J[k].append(gi.reshape(-1))
    save(name, meta=np.array([ns, nc, T, B]), beta=np.array([beta]), C=C, c=c, x_init=x_init, F=F, f=f,
         x=npy(x), u=npy(u), **{"J_" + k: torch.stack(v).numpy() for k, v in J.items()})


# --------------------------------------------------------------------------
# pnqp, get_traj / get_cost
# --------------------------------------------------------------------------
def pnqp_case(name, B, n, dtype, seed, warm):
    npr.seed(seed)
    H = npr.randn(B, n, n)
    H = np.matmul(H.transpose(0, 2, 1), H)
    q = npr.randn(B, n)
    lower = -npr.random((B, n))
    upper = npr.random((B, n))
    x0 = 0.5 * npr.randn(B, n) if warm else None
    tt = lambda a: None if a is None else torch.tensor(a).to(dtype)
    H_, q_, lo_, hi_, x0_ = map(tt, (H, q, lower, upper, x0))
    (xb, _, Ifb, itb), _ = quiet(ref_pnqp.pnqp, H_, q_, lo_, hi_, x_init=x0_)
    xs, Ifs, its = [], [], []
    for b in range(B):
        (x1, _, If1, it1), _ = quiet(ref_pnqp.pnqp, H_[b:b + 1], q_[b:b + 1], lo_[b:b + 1], hi_[b:b + 1],
                                     x_init=None if x0_ is None else x0_[b:b + 1])
        xs.append(x1); Ifs.append(If1); its.append(it1)
    save(name, H=npy(H_), q=npy(q_), lower=npy(lo_), upper=npy(hi_), x0=npy(x0_),
         x_batch=npy(xb), If_batch=npy(Ifb), iters_batch=np.array([itb]),
         x_pp=npy(torch.cat(xs)), If_pp=npy(torch.cat(Ifs)), iters_pp=np.array(its))


def traj_cost_case():
    p = make_problem(4, 2, 7, 3, torch.float64, 5)
    x = ref_util.get_traj(7, p["u"], p["x_init"], LinDx(p["F"], p["f"]))
    cost = ref_util.get_cost(7, p["u"], QuadCost(p["C"], p["c"]), LinDx(p["F"], p["f"]), x_init=p["x_init"])
    save("traj_cost", **{k: npy(v) for k, v in p.items()}, x=npy(x), cost=npy(cost))


if __name__ == "__main__":
    f64, f32 = torch.float64, torch.float32
    only = set(a for a in sys.argv[1:] if not a.startswith("--only="))
    for a in sys.argv[1:]:
        if a.startswith("--only="):
            ONLY = set(a[len("--only="):].split(","))
            only = {"step"}
    if only and "step" not in only:
        step_case = lambda *a, **k: None
    if only and "mpc" not in only:
        gen_mpc_cases = lambda: None
    _grad_case = grad_case
    if only and "grad" not in only:
        grad_case = jacobian_case = lambda *a, **k: None
    if only and "misc" not in only:
        pnqp_case = lambda *a, **k: None
        traj_cost_case = lambda: None
    # ---- NNDynamics as the dynamics (`nn` alone regenerates just these) -----
    if not only or "nn" in only:
        nn_case("nn_sigmoid_f64", 71, 4, 2, [24], "sigmoid", True, 8, 5, 1.0)
        nn_case("nn_relu2_f64", 72, 6, 3, [20, 12], "relu", True, 7, 4, None, w_scale=1.5)
        nn_case("nn_headline_f64", 73, 12, 4, [100], "sigmoid", True, 10, 6, 0.5)
        nn_case("nn_nopass_f64", 74, 3, 1, [16, 16, 8], "sigmoid", False, 6, 4, 0.8, decay=0.5, max_ls=4)
    if only == {"nn"}:
        sys.exit(0)
    # ---- single LQR steps -------------------------------------------------
    step_case("step_cfg1_f64", 4, 2, 10, 8, f64, 11, bounds="tensor")
    step_case("step_cfg1_f32", 4, 2, 10, 8, f32, 11, bounds="tensor")
    step_case("step_unbounded_f64", 3, 4, 5, 2, f64, 12, bounds=None)
    step_case("step_unbounded_nof_f64", 3, 4, 6, 3, f64, 13, with_f=False, bounds=None)
    step_case("step_ns_unbounded_f32", 12, 4, 50, 4, f32, 14, bounds=None)
    step_case("step_ns_bounded_f32", 12, 4, 50, 4, f32, 14, bounds=1.0)
    step_case("step_ns_bounded_f64", 12, 4, 20, 3, f64, 15, bounds=1.0)
    step_case("step_nc1_scalar_f64", 3, 1, 20, 4, f64, 16, bounds=2.0, decay=0.2, max_ls=5)
    step_case("step_nc1_unbounded_f32", 5, 1, 25, 4, f32, 17, bounds=None, decay=0.5, max_ls=2)
    step_case("step_nc1_bounded_f32", 5, 1, 25, 4, f32, 17, bounds=0.5, decay=0.5, max_ls=2, u_scale=1.0)
    step_case("step_delta_f64", 3, 4, 5, 2, f64, 18, bounds="tensor", delta_u=0.1)
    step_case("step_masked_f64", 3, 2, 6, 4, f64, 19, bounds=None, mask_seed=3)
    step_case("step_masked_nc1_f64", 3, 1, 6, 6, f64, 20, bounds=None, mask_seed=4)
    step_case("step_masked_nc4_f32", 12, 4, 12, 3, f32, 21, bounds=None, mask_seed=5)
    step_case("step_cfg5_f32", 32, 8, 8, 2, f32, 22, bounds=None)
    step_case("step_cfg5_bounded_f64", 32, 8, 6, 2, f64, 23, bounds=0.5)
    step_case("step_linesearch_f64", 4, 2, 8, 6, f64, 24, bounds=0.3, u_scale=2.0, decay=0.5, max_ls=4)
    step_case("step_backtrack_a_f64", 4, 2, 8, 6, f64, 45, bounds=0.3, u_scale=2.0, decay=0.5, max_ls=4, indef=6.0)
    step_case("step_backtrack_b_f64", 4, 2, 8, 6, f64, 40, bounds=0.3, u_scale=2.0, decay=0.5, max_ls=4, indef=6.0)
    # ---- a C that is not symmetric (round 3): some problems of the batch only, every kernel family
    step_case("step_asym_ns_f32", 12, 4, 12, 6, f32, 61, bounds=None, asym=(1, 4, 5))
    step_case("step_asym_ns_bounded_f32", 12, 4, 12, 6, f32, 62, bounds=1.0, asym=(0, 3))
    step_case("step_asym_ns_masked_f32", 12, 4, 10, 5, f32, 63, bounds=None, mask_seed=7, asym=(2,))
    step_case("step_asym_tiny_f32", 12, 4, 8, 4, f32, 64, bounds=None, asym=(0, 2), asym_scale=1e-7)   # rounding-level: symmetric for all purposes
    step_case("step_asym_cfg5_f32", 32, 8, 6, 3, f32, 65, bounds=None, asym=(1,))
    step_case("step_asym_small_f64", 4, 2, 8, 3, f64, 66, bounds=None, asym=(0, 1, 2))
    step_case("step_asym_nc1_f32", 5, 1, 10, 4, f32, 67, bounds=0.5, decay=0.5, max_ls=2, asym=(1, 3))
    step_case("step_asym_odd_f32", 7, 3, 9, 3, f32, 68, bounds=None, asym=(2,))
    # ---- rank-deficient Quu: a control that enters neither cost nor dynamics (pinverse, mpc/lqr_step.py:88-94)
    step_case("step_singular_ns_f32", 12, 4, 12, 5, f32, 71, bounds=None, zero_ctrl=((1, 2), (3, 0)))
    step_case("step_singular_cfg5_f64", 32, 8, 6, 3, f64, 72, bounds=None, zero_ctrl=((0, 5),))     # (float32: the reference's own 8x8 pinverse is noise there)
    step_case("step_singular_small_f64", 4, 2, 8, 3, f64, 73, bounds=None, zero_ctrl=((1, 1),))
    step_case("step_singular_odd_f32", 7, 3, 9, 3, f32, 74, bounds=None, zero_ctrl=((2, 0),))
    tie_case("ties_tight_f32")
    du_norm_case("mpc_du_norm_B8_f64")
    if ONLY is not None:
        _grad_case("grad_asym_ns_f32", 12, 4, 8, 4, f32, 36, None, asym=(1, 2))
        _grad_case("grad_asym_small_f64", 4, 2, 6, 3, f64, 37, 0.4, asym=(0, 2))
        _grad_case("grad_cfg5_unconstrained_f32", 32, 8, 7, 3, f32, 38, None)
        _grad_case("grad_cfg5_constrained_f32", 32, 8, 6, 3, f32, 39, 0.3)
        _grad_case("grad_asym_cfg5_f32", 32, 8, 6, 3, f32, 40, None, asym=(1,))
        _grad_case("grad_asym_20_4_f32", 20, 4, 7, 3, f32, 41, 0.5, asym=(0, 2))
        sys.exit(0)
    # ---- full solves --------------------------------------------------------
    if not only or "env" in only:
        env_case("ilqr_pendulum_f64", "pendulum", 20, 4, 8, 51)
        env_case("ilqr_cartpole_f64", "cartpole", 25, 4, 8, 52)
        slew_case("mpc_slew_nn_f64", 0)
        slew_case("mpc_slew_nn_prev_f64", 3, T=5, B=3, gamma=0.5, prev=True)
        module_cost_case("mpc_module_cost_f64", 7)
        module_cost_case("mpc_module_cost_wide_f64", 8, ns=4, nc=1, T=6, B=4, bound=5.0)
        env_lin_case("env_pendulum_f64", "pendulum", 12, 6, 61)
        env_lin_case("env_pendulum_full_f64", "pendulum", 12, 6, 62, simple=False,
                     params=(9.0, 1.2, 0.8, 0.3, 0.2))
        env_lin_case("env_cartpole_f64", "cartpole", 12, 6, 63)
    if only == {"env"}:
        sys.exit(0)
    gen_mpc_cases()
    # ---- backward -------------------------------------------------------------
    grad_case("grad_unconstrained_f64", 4, 2, 6, 4, f64, 31, None)
    grad_case("grad_constrained_f64", 4, 2, 6, 4, f64, 31, 0.4)
    grad_case("grad_constrained_nof_f64", 3, 2, 5, 3, f64, 32, 0.3, with_f=False)
    grad_case("grad_nc1_constrained_f64", 3, 1, 6, 4, f64, 33, 0.3)
    grad_case("grad_ns_constrained_f32", 12, 4, 10, 3, f32, 34, 0.5)
    grad_case("grad_ns_unconstrained_f64", 12, 4, 10, 2, f64, 35, None)
    grad_case("grad_asym_ns_f32", 12, 4, 8, 4, f32, 36, None, asym=(1, 2))
    grad_case("grad_asym_small_f64", 4, 2, 6, 3, f64, 37, 0.4, asym=(0, 2))
    # config 5's shape (round 3: the backward fused into its nested step, lqr_mfma40_body.h)
    grad_case("grad_cfg5_unconstrained_f32", 32, 8, 7, 3, f32, 38, None)
    grad_case("grad_cfg5_constrained_f32", 32, 8, 6, 3, f32, 39, 0.3)
    # round 4: a C that is not symmetric through the backward of the shapes kkt_wave.hip takes (its costate
    # recursion read C by columns)
    grad_case("grad_asym_cfg5_f32", 32, 8, 6, 3, f32, 40, None, asym=(1,))
    grad_case("grad_asym_20_4_f32", 20, 4, 7, 3, f32, 41, 0.5, asym=(0, 2))
    jacobian_case("jac_unconstrained", 100.0)
    jacobian_case("jac_constrained", 0.5)
    # ---- pnqp / traj --------------------------------------------------------
    pnqp_case("pnqp_n100_f64", 2, 100, f64, 1, warm=False)
    pnqp_case("pnqp_n4_warm_f64", 16, 4, f64, 2, warm=True)
    pnqp_case("pnqp_n4_cold_f32", 16, 4, f32, 3, warm=False)
    pnqp_case("pnqp_n1_f64", 8, 1, f64, 4, warm=False)
    pnqp_case("pnqp_n8_warm_f32", 8, 8, f32, 5, warm=True)
    traj_cost_case()
