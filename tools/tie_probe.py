#!/usr/bin/env python3
"""Round 5 diagnostic: which problems of a box-constrained batch leave the tolerance against the float64 oracle, and are they
active-set ties (a QP minimiser on its bound to within rounding: the clamped rows of K differ from the float64 run's)?
    python tools/r05_tie_probe.py ns nc T B seed [c_symmetric]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import bench
from mpc import _native
from mpc._native import StepOptions
from oracle import lqr_oracle as O
ns, nc, T, B, seed = (int(v) for v in sys.argv[1:6])
sym = len(sys.argv) < 7 or sys.argv[6] != "0"
be = _native.HipBackend()
p = bench.make_problem(ns, nc, T, B, torch.float32, "cuda:0", seed=seed, u_scale=0.3, clamp=1.0)
h = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in p.items()}
o = O.lqr_step(h["x_init"], h["C"], h["c"], h["F"], h["f"], h["cur_x"], h["cur_u"], -1.0, 1.0, lockstep=False, nthreads=O.max_threads(), return_gains=True)
for vouch in (True, False):
    r = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"],
                    StepOptions(u_lower=-1.0, u_upper=1.0, nominal_on_dynamics=vouch, c_symmetric=sym and vouch), want_gains=True)
    torch.cuda.synchronize()
    nu, K = r["new_u"].cpu().numpy().astype(np.float64), r["K"].cpu().numpy()
    eu = (np.abs(nu - o["new_u"]) / (1e-4 + 1e-3 * np.abs(o["new_u"]))).max(axis=(0, 2))
    bad = np.nonzero(eu > 1)[0]
    print("vouch", vouch, "bad problems", bad.tolist(), "of", B, "status", np.unique(r["status"].cpu().numpy()).tolist())
    for b in bad[:6]:
        zr, zo = (K[:, b] == 0).all(-1), (o["K"][:, b] == 0).all(-1)
        d = np.argwhere(zr != zo)
        t0 = int(d[:, 0].max()) if len(d) else -1
        print("  problem", int(b), "err/tol %.1f" % eu[b], "alpha", float(r["alphas"][b]), o["alphas"][b], "cost", float(r["costs"][b]), o["costs"][b], "old", o["old_costs"][b],
              "| clamped sets differ at (t, a):", d[-4:].tolist())
        if t0 >= 0:
            kk = o["k"][t0, b]; lb = -1.0 - h["cur_u"][t0, b]; ub = 1.0 - h["cur_u"][t0, b]
            print("     at t=%d oracle k %s lb %s ub %s gpu k %s" % (t0, np.round(kk, 6), np.round(lb, 6), np.round(ub, 6), np.round(r["k"][t0, b].cpu().numpy(), 6)))
