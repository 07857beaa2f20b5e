#!/usr/bin/env python3
"""Randomised parity of the NNDynamics kernels against oracle/env_oracle.py: random shapes (n_state 1..16, n_ctrl 1..8, 0..3
hidden layers of 1..300 units, every activation, passthrough on / off, ragged batches, T 1..12), trajectory, linearisation
and the line-searched rollout with random bounds.  One JSON line per case; exits non-zero on a violation."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mpc import _native
from mpc._native import StepOptions, MlpSpec
from oracle import env_oracle as E
from oracle import lqr_oracle as O
be = _native.HipBackend()
DEV = "cuda:0"
f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(DEV)
host = lambda t: t.detach().cpu().numpy().astype(np.float64)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for case in range(n_cases):
    rng = np.random.RandomState(1000 + case)
    ns, nc = int(rng.randint(1, 17)), int(rng.randint(1, 9))
    hidden = [int(rng.randint(1, 301)) for _ in range(rng.randint(0, 4))]
    act = ("sigmoid", "relu", "elu")[rng.randint(3)]
    passthrough = bool(rng.randint(2))
    T, B = int(rng.randint(1, 13)), int(rng.randint(1, 200))
    sizes = [ns + nc] + hidden + [ns]
    net = E.Mlp([0.8 * rng.uniform(-1, 1, (o, i)) / np.sqrt(i) for i, o in zip(sizes, sizes[1:])],
                [rng.uniform(-1, 1, o) / np.sqrt(i) for i, o in zip(sizes, sizes[1:])], act, passthrough)
    sp = MlpSpec([f32(W) for W in net.Ws], [f32(b) for b in net.bs], act, passthrough)
    n = ns + nc
    x0, u0 = rng.randn(B, ns), 0.3 * rng.randn(T, B, nc)
    bound = None if rng.randint(3) == 0 else float(rng.uniform(0.2, 1.0))
    if bound is not None:
        u0 = np.clip(u0, -bound, bound)
    A = rng.randn(T, B, n, n)
    C = np.einsum("tbki,tbkj->tbij", A, A) + 0.1 * np.eye(n)
    c = rng.randn(T, B, n)
    xs = E.traj(E.MLP, x0, u0, net)
    scale = 1.0 + np.abs(xs).max()
    row = dict(case=case, ns=ns, nc=nc, hidden=hidden, act=act, passthrough=passthrough, T=T, B=B, bound=bound)
    xk, ck = be.mlp_traj_cost(f32(x0), f32(u0), sp, C=f32(C), c=f32(c))
    row["traj_err"] = float(np.abs(host(xk) - xs).max() / scale)
    old = E.quad_cost(C, c, xs, u0)
    row["cost_err"] = float((np.abs(host(ck) - old) / (1 + np.abs(old))).max())
    ok = row["traj_err"] < 3e-4 and row["cost_err"] < 2e-3
    if T > 1:
        Fl, fl = E.linearize(E.MLP, xs[:-1].reshape(-1, ns), u0[:-1].reshape(-1, nc), net)
        Fk, fk = be.mlp_linearize(sp, f32(xs[:-1].reshape(-1, ns)), f32(u0[:-1].reshape(-1, nc)))
        row["F_err"] = float(np.abs(host(Fk) - Fl).max() / (1 + np.abs(Fl).max()))
        row["f_err"] = float(np.abs(host(fk) - fl).max() / scale)
        ok = ok and row["F_err"] < 3e-4 and row["f_err"] < 5e-4
        Fl, fl = Fl.reshape(T - 1, B, ns, n), fl.reshape(T - 1, B, ns)
    else:
        Fl, fl = np.zeros((0, B, ns, n)), np.zeros((0, B, ns))
    lo, hi = (None, None) if bound is None else (-bound, bound)
    o = O.lqr_step(x0, C, c, Fl, fl if T > 1 else None, xs, u0, lo, hi, linesearch_decay=0.2, max_linesearch_iter=5,
                   lockstep=False, return_gains=True)
    nx, nu, costs, full, alphas, trials, old2 = E.rollout_batched(E.MLP, net, x0, C, c, o["K"], o["k"], xs, u0, lo, hi, 0.2, 5)
    r = be.mlp_rollout(f32(x0), f32(C), f32(c), f32(o["K"]), f32(o["k"]), f32(xs), f32(u0), f32(old2),
                       StepOptions(u_lower=lo, u_upper=hi, linesearch_decay=0.2, max_linesearch_iter=5), sp)
    torch.cuda.synchronize()
    same = np.isclose(host(r["alphas"]), alphas, rtol=1e-5)
    row["alpha_ties"] = int((~same).sum())
    row["u_err"] = float(np.abs(host(r["new_u"]) - nu)[:, same].max() / scale) if same.any() else 0.0
    row["x_err"] = float(np.abs(host(r["new_x"]) - nx)[:, same].max() / scale) if same.any() else 0.0
    ok = ok and row["u_err"] < 1e-3 and row["x_err"] < 1e-3 and row["alpha_ties"] <= max(1, B // 50)
    row["ok"] = bool(ok)
    bad += 0 if ok else 1
    print(json.dumps(row), flush=True)
print("violations", bad)
sys.exit(1 if bad else 0)
