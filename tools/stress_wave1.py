#!/usr/bin/env python3
"""Randomised parity stress of the row-per-problem kernel (impl 6) over its whole domain: n_state 1..6, T 1..70 (several rounds
of its timestep-parallel phase), ragged batches, every bound mode, delta_u, u_zero_I, 1..16 line-search trials, with and without
f, the three shipped simulators with given and in-kernel Jacobians -- against the float64 oracle and against the lane-per-problem
kernel (impl 4) on the same float32 inputs.  One line per case; exits non-zero on a violation.
Problems whose accepted step size differs from the oracle's (a float32 tie of two trial costs) are counted, must stay rare, and are
left out of the entry-wise comparison -- the reference's line search is discontinuous there."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
from mpc import _native
from mpc._native import StepOptions, EnvSpec
from oracle import lqr_oracle as O
be = _native.HipBackend()
DEV = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(2026)
bad = 0
for case in range(N):
    ns = int(rng.integers(1, 7)); n = ns + 1
    T = int(rng.choice([1, 2, 3, 7, 16, 17, 25, 33, 50, 70])); B = int(rng.choice([1, 3, 4, 5, 37, 64, 130, 257]))
    max_ls = int(rng.choice([1, 2, 3, 5, 8, 10, 16])); decay = float(rng.choice([0.2, 0.5]))
    mode = str(rng.choice(["free", "scalar", "tensor", "delta", "masked"]))
    A = rng.standard_normal((T, B, n, n))
    C = np.einsum("tbji,tbjk->tbik", A, A) + 0.1 * np.eye(n)
    if rng.random() < 0.5:
        C[:, :, :ns, :ns] -= 2.0 * np.eye(ns)                  # an indefinite state block: the line search backtracks
    if rng.random() < 0.5:
        C = C + 0.05 * rng.standard_normal(C.shape)            # not symmetric: both kernels use C as given
    c = rng.standard_normal((T, B, n))
    F = np.concatenate((np.eye(ns) + 0.3 * rng.standard_normal((max(T - 1, 0), B, ns, ns)) / np.sqrt(ns), rng.standard_normal((max(T - 1, 0), B, ns, 1))), 3)
    f = None if rng.random() < 0.3 else 0.1 * rng.standard_normal((max(T - 1, 0), B, ns))
    x_init = rng.standard_normal((B, ns))
    cur_u = np.clip(0.3 * rng.standard_normal((T, B, 1)), -0.4, 0.4)
    f32 = lambda a: None if a is None else a.astype(np.float32)
    x_init, C, c, F, f, cur_u = (f32(a) for a in (x_init, C, c, F, f, cur_u))
    d64 = lambda a: None if a is None else a.astype(np.float64)
    cur_x = f32(O.traj_cost(d64(x_init), d64(cur_u), d64(F), d64(f))[0])
    kw, okw = dict(linesearch_decay=decay, max_linesearch_iter=max_ls), {}
    if mode == "scalar":
        okw = dict(u_lower=-0.5, u_upper=0.5)
    elif mode == "tensor":
        lo, hi = f32(-0.5 - rng.random((T, B, 1))), f32(0.5 + rng.random((T, B, 1)))
        okw = dict(u_lower=lo, u_upper=hi)
    elif mode == "delta":
        okw = dict(u_lower=-0.5, u_upper=0.5, delta_u=0.05)
    elif mode == "masked":
        okw = dict(u_zero_I=rng.random((T, B, 1)) < 0.3)
    o = O.lqr_step(d64(x_init), d64(C), d64(c), d64(F), d64(f), d64(cur_x), d64(cur_u),
                   **{k: (d64(v) if isinstance(v, np.ndarray) and v.dtype.kind == "f" else v) for k, v in okw.items()},
                   lockstep=False, nthreads=O.max_threads(), **kw)
    tk = {k: (torch.from_numpy(v).to(DEV) if isinstance(v, np.ndarray) else v) for k, v in okw.items()}
    opts = StepOptions(**tk, **kw)
    dev = lambda a: None if a is None else torch.from_numpy(a).to(DEV)
    args = [dev(a) for a in (x_init, C, c, F, f, cur_x, cur_u)]
    lds_floats = T * (n * n + 3 * n + ns * n + (ns if f is not None else 0) + (2 if mode in ("scalar", "tensor", "delta") else 0)
                      + (1 if mode == "masked" else 0) + max_ls * n)                     # wave1::layout
    if lds_floats * 16 > 150 * 1024:
        print(json.dumps(dict(case=case, ns=ns, T=T, B=B, skipped="too large for four problems in LDS")))
        continue
    r = be.lqr_step(*args, opts, impl=6)
    t = be.lqr_step(*args, opts, impl=4)
    torch.cuda.synchronize()
    h = lambda v: v.cpu().numpy()
    same = np.isclose(h(r["alphas"]), o["alphas"], rtol=1e-5) & np.isclose(h(t["alphas"]), o["alphas"], rtol=1e-5)
    flips6 = int((~np.isclose(h(r["alphas"]), o["alphas"], rtol=1e-5)).sum())
    flips4 = int((~np.isclose(h(t["alphas"]), o["alphas"], rtol=1e-5)).sum())
    line = dict(case=case, ns=ns, T=T, B=B, max_ls=max_ls, mode=mode, f=f is not None, alpha_flips=flips6, alpha_flips_lane_kernel=flips4)
    # float32 against float64 on random problems, some of them badly conditioned (an indefinite state block over 70 timesteps
    # amplifies rounding by orders of magnitude in ANY float32 evaluation -- the lane-per-problem kernel is off by 0.7 on one of
    # them): a case counts against the kernel when entries are out of tolerance AND its worst error is more than ten times
    # the lane-per-problem kernel's worst error on the same inputs
    over = 0
    for k in ("new_x", "new_u", "costs", "old_costs"):
        sel = (lambda v: v[:, same]) if k in ("new_x", "new_u") else (lambda v: v[same])
        a, b, w = sel(h(r[k])), sel(h(t[k])), sel(o[k])
        e6, e4 = np.abs(a - w), np.abs(b - w)
        n_out = int((e6 > 5e-4 + 2e-3 * np.abs(w)).sum())
        m6, m4 = (float(e6.max()), float(e4.max())) if same.any() else (0.0, 0.0)
        if n_out and m6 > 10 * max(m4, 1e-4):
            over += n_out
        if k in ("new_x", "new_u"):
            line["max_err_" + k], line["max_err_lane_kernel_" + k] = m6, m4
    st = h(r["status"])
    line.update(over_tol=over, nonfinite=int((st & 2 != 0).sum()), status_differs=int((st != h(t["status"])).sum()))
    # (a deep line search on a non-convex problem ends where two trial costs tie in float32: either kernel may fall either way)
    viol = over > 0 or flips6 > flips4 + max(2, B // 20) or line["nonfinite"] or not np.isfinite(h(r["new_x"])).all()
    bad += bool(viol)
    print(json.dumps(line) + ("   <-- VIOLATION" if viol else ""), flush=True)
# the shipped simulators, at iLQR-like nominals
from tools.bench_ilqr_env import problem
for kind, B, T in (("pendulum", 333, 20), ("cartpole", 1001, 25), ("pendulum", 5, 64), ("cartpole", 2, 3)):
    dx, plain, x0, Q, pp = problem(kind, B, T)
    for lin in (True, False):
        g = torch.Generator().manual_seed(B)
        u = (0.5 * torch.randn(T, B, 1, generator=g)).to(DEV)
        env = dx.native_env()
        x, _ = be.env_traj_cost(x0, u, env)
        Fl, fl = be.env_linearize(env, x[:-1].reshape(-1, dx.n_state), u[:-1].reshape(-1, 1))
        Fl, fl = Fl.view(T - 1, B, dx.n_state, -1), fl.view(T - 1, B, -1)
        env.linearize = lin
        o6 = StepOptions(u_lower=dx.lower, u_upper=dx.upper, linesearch_decay=dx.linesearch_decay, max_linesearch_iter=dx.max_linesearch_iter, true_dynamics=env)
        a = (x0, Q, pp, None if lin else Fl, None if lin else fl, x, u)
        r = be.lqr_step(*a, o6, impl=6); t = be.lqr_step(*a, o6, impl=4)
        torch.cuda.synchronize()
        same = torch.isclose(r["alphas"], t["alphas"], rtol=1e-5)
        ex = float((r["new_x"] - t["new_x"])[:, same].abs().max()); eu = float((r["new_u"] - t["new_u"])[:, same].abs().max())
        viol = ex > 2e-3 or eu > 2e-3 or int((~same).sum()) > max(2, B // 20)
        bad += bool(viol)
        print(json.dumps(dict(simulator=kind, B=B, T=T, in_kernel_jacobian=lin, alpha_flips=int((~same).sum()), max_diff_x=ex, max_diff_u=eu)) + ("   <-- VIOLATION" if viol else ""), flush=True)
print("violations:", bad)
sys.exit(1 if bad else 0)
