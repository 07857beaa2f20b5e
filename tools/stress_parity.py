#!/usr/bin/env python3
"""Randomised parity stress at the headline shape: every kernel against the float64 oracle over several
seeds and option sets.  Prints one line per (case, seed, impl); exits non-zero on a violation.

Two kinds of problems are compared on their own terms, not entry by entry, because the reference algorithm
itself is discontinuous there: a line search whose trial cost ties with the nominal cost (other alpha), and a
box QP whose minimiser sits on a bound to within rounding (the component is "clamped" or "free" by the sign
of a ~1e-7 gradient, which zeroes or keeps a row of K).  Both are counted and must stay rare."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import bench
from mpc import _native
from mpc._native import StepOptions
from oracle import lqr_oracle as O
be = _native.HipBackend()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
NS, NC, TT = (int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (12, 4, 50)
DT = torch.float64 if "f64" in sys.argv else torch.float32       # (round 5: "f64" as a last argument stresses the float64 kernels)
RTOL, ATOL = (1e-6, 1e-7) if DT == torch.float64 else (1e-3, 1e-4)   # (float64: the box QP's own stopping rule, |dx| < 1e-4, squared)
IMPLS = [i for i in (1, 2, 3, 4, 5, 6, 7) if be.impl_supported(NS, NC, DT, i)]
if len(sys.argv) > 3 and sys.argv[3] != "f64":                      # e.g. "3,7": only these kernels (the generic one is slow at large batches)
    IMPLS = [i for i in IMPLS if str(i) in sys.argv[3].split(",")]
bad = 0
for case in ("unbounded", "bounded", "tensor_bounds", "delta_u", "tight"):
    for seed in range(4):
        u_scale, clamp = (0.0, None) if case == "unbounded" else (0.3, 1.0)
        if case == "tight":
            u_scale, clamp = 0.2, 0.3
        p = bench.make_problem(NS, NC, TT, B, DT, "cuda:0", seed=100 + seed, u_scale=u_scale, clamp=clamp)
        if DT == torch.float64:       # (C = A'A out of a float32 product is symmetric to 1e-7 only: the reference uses C as given)
            p["C"] = 0.5 * (p["C"] + p["C"].transpose(2, 3))
        h = {k: v.cpu().numpy().astype(np.float64) for k, v in p.items()}
        kw = {}
        if case == "bounded":
            kw = dict(u_lower=-1.0, u_upper=1.0)
        elif case == "tight":
            kw = dict(u_lower=-0.3, u_upper=0.3)
        elif case == "tensor_bounds":
            g = torch.Generator().manual_seed(seed)
            lo = (-1.0 - torch.rand(TT, B, NC, generator=g)).to(DT).cuda(); hi = (1.0 + torch.rand(TT, B, NC, generator=g)).to(DT).cuda()
            kw = dict(u_lower=lo, u_upper=hi)
        elif case == "delta_u":
            kw = dict(u_lower=-1.0, u_upper=1.0, delta_u=0.25)
        okw = {k: (v.cpu().numpy().astype(np.float64) if torch.is_tensor(v) else v) for k, v in kw.items()}
        o = O.lqr_step(h["x_init"], h["C"], h["c"], h["F"], h["f"], h["cur_x"], h["cur_u"], okw.get("u_lower"), okw.get("u_upper"),
                       delta_u=okw.get("delta_u"), lockstep=False, nthreads=O.max_threads(), return_gains=True)
        opat = (o["K"] == 0).all(axis=-1)
        for impl in IMPLS:
            r = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], StepOptions(**kw), impl=impl,
                            want_gains=True)
            torch.cuda.synchronize()
            ties = ((r["K"].cpu().numpy() == 0).all(axis=-1) != opat).any(axis=(0, 2))
            same = np.isclose(r["alphas"].cpu().numpy(), o["alphas"], rtol=1e-5) & ~ties
            ex = np.abs(r["new_x"].cpu().numpy() - o["new_x"])[:, same]
            eu = np.abs(r["new_u"].cpu().numpy() - o["new_u"])[:, same]
            lim_x = ATOL + RTOL * np.abs(o["new_x"][:, same]); lim_u = ATOL + RTOL * np.abs(o["new_u"][:, same])
            st = r["status"].cpu().numpy()
            line = dict(case=case, seed=seed, impl=impl, alpha_flips_or_ties=int((~same).sum()), active_set_ties=int(ties.sum()), max_err_x=float(ex.max()), max_err_u=float(eu.max()),
                        over_tol=int((ex > lim_x).sum() + (eu > lim_u).sum()), unconverged=int((st & 1).sum()), nonfinite=int((st & 2 != 0).sum()))
            viol = line["over_tol"] > 0 or line["alpha_flips_or_ties"] > max(2, B // 500) or line["nonfinite"]
            bad += bool(viol)
            print(json.dumps(line) + ("   <-- VIOLATION" if viol else ""))
sys.exit(1 if bad else 0)
