#!/usr/bin/env python3
"""No GPU: the fused kernels' bodies on the CPU wavefront emulator (tests/emu) against the float64 oracle over RANDOM option sets --
batch (ragged waves), horizon, bounds (none / scalar / tensor), delta_u, u_zero_I, f on / off, line-search depth and decay, promises,
qp_start -- on convex problems, where parity is exact up to float32 rounding.  One line per violation; exits non-zero if any.
    python tools/emu_fuzz.py [cases [seed [kernel,...]]]        kernels: dpp16 dpp16_ring2 mfma16 mfma40"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd"))
from oracle import lqr_oracle as O
import emu_backend as emu

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
kernels = sys.argv[3].split(",") if len(sys.argv) > 3 else ["dpp16", "dpp16_ring2", "mfma16"]
bad = 0
t0 = time.time()
only = os.environ.get("FUZZ_ONLY")                  # FUZZ_ONLY=<case>[:kernel]: that case alone (with another kernel on the same problem)
for case in range(cases):
    if only and case != int(only.split(":")[0]):
        continue
    rng = np.random.default_rng(seed0 * 100003 + case)
    kernel = kernels[case % len(kernels)]
    ns, nc = (32, 8) if kernel.startswith("mfma40") else (12, 4)
    f64 = kernel == "mfma16_f64"                       # the float64 instantiation of the one-problem-per-wave body
    if kernel in ("mfma16", "mfma16_f64") and rng.random() < 0.5:
        ns, nc = int(rng.integers(1, 13)), int(rng.integers(1, 5))
    if kernel == "mfma40_pad":                        # the padded instantiation of the 32/8 body: dword or 16-byte gathers
        if rng.random() < 0.5:
            ns, nc, kernel = int(rng.integers(1, 33)), int(rng.integers(1, 9)), "mfma40_pad4"
        else:
            ns, nc, kernel = 4 * int(rng.integers(1, 9)), 4 * int(rng.integers(1, 3)), "mfma40_pad16"
    n = ns + nc
    long_T = os.environ.get("FUZZ_LONG_T")            # horizons across the register-resident gains' limit (64) and several ring turns
    T = int(rng.choice([1, 2, 3, 5, 8, 13] + ([40, 66] if long_T else []))) if ns > 12 else int(rng.choice([1, 2, 3, 4, 5, 7, 9, 12, 17] + ([33, 63, 64, 65, 70] if long_T else [])))
    B = int(rng.choice([1, 2, 3])) if ns > 12 else int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9]))
    A = rng.standard_normal((T, B, n, n))
    C = np.einsum("tbji,tbjk->tbik", A, A) + 0.05 * np.eye(n)
    c = rng.standard_normal((T, B, n))
    # (a long horizon on dynamics that grow 1.4x a step is a nominal 1e10 times its start: every float32 code, the reference's included,
    #  is noise there -- long horizons keep the benchmark recipe's 0.2)
    fscale = float(rng.choice([0.1, 0.2, 0.4])) if T <= 20 else float(rng.choice([0.1, 0.2]))
    F = np.concatenate((np.eye(ns) + fscale * rng.standard_normal((max(T - 1, 0), B, ns, ns)) / np.sqrt(ns),
                        rng.standard_normal((max(T - 1, 0), B, ns, nc)) / np.sqrt(ns)), 3)
    f = 0.1 * rng.standard_normal((max(T - 1, 0), B, ns)) if (rng.random() < 0.7 and T > 1) else None
    x_init = rng.standard_normal((B, ns))
    bnd = float(rng.choice([0.2, 0.4, 1.0]))
    cur_u = np.clip(float(rng.choice([0.0, 0.3, 0.6])) * rng.standard_normal((T, B, nc)), -bnd, bnd)
    cur_x, _ = O.traj_cost(x_init, cur_u, F, f)
    kw = dict(x_init=x_init, C=C, c=c, F=F, f=f, cur_x=cur_x, cur_u=cur_u,
              linesearch_decay=float(rng.choice([0.2, 0.5, 0.8])), max_linesearch_iter=int(rng.choice([1, 2, 3, 5, 10])))
    mode = rng.choice(["none", "scalar", "tensor", "mask"], p=[0.2, 0.35, 0.3, 0.15])
    if mode == "scalar":
        kw.update(u_lower=-bnd, u_upper=bnd)
    elif mode == "tensor":
        kw.update(u_lower=-bnd - 0.3 * rng.random((T, B, nc)), u_upper=bnd + 0.3 * rng.random((T, B, nc)))
    elif mode == "mask":
        kw.update(u_zero_I=(rng.random((T, B, nc)) < 0.3))
        cur_u = np.where(kw["u_zero_I"], 0.0, cur_u); cur_x, _ = O.traj_cost(x_init, cur_u, F, f); kw.update(cur_u=cur_u, cur_x=cur_x)
    if mode in ("scalar", "tensor") and rng.random() < 0.25:
        kw.update(delta_u=float(rng.choice([0.05, 0.2, 1.0])))
    o = O.lqr_step(lockstep=False, **kw)
    ekw = dict(kernel="mfma16" if f64 else kernel, **({"dtype": np.float64} if f64 else {}), dma_late=bool(rng.integers(0, 2)), nominal_on_dynamics=bool(rng.integers(0, 2)), c_symmetric=bool(rng.integers(0, 2)))
    if kernel == "mfma16" and (ns, nc) == (12, 4) and not f64:
        ekw["force_general"] = bool(rng.integers(0, 2))
    if mode in ("scalar", "tensor") and kernel in ("dpp16", "dpp16_ring2", "mfma40", "mfma40_ring2") and rng.random() < 0.3:
        ekw["qp_start"] = rng.standard_normal((T, B, nc)) if rng.random() < 0.5 else np.zeros((1, 1, nc))
    if only and ":" in only:
        ekw["kernel"] = only.split(":")[1]
        if len(only.split(":")) > 2:
            ekw["nominal_on_dynamics"] = only.split(":")[2] == "vouched"
    try:
        r = emu.lqr_step(**ekw, **kw)
    except AssertionError as e:
        print("CASE %d %s: emulator refused (%s) -- %s" % (case, kernel, e, {k: (v if np.isscalar(v) or isinstance(v, bool) else "array") for k, v in ekw.items()}))
        continue
    # ties of the line search (a trial cost within rounding of the nominal's) are named, not compared
    flip = ~np.isclose(r["alphas"], o["alphas"], rtol=1e-6)
    tie = np.abs(o["costs"] - o["old_costs"]) <= 1e-5 * (1 + np.abs(o["old_costs"]))
    k = ~flip
    scale = 1 + np.abs(o["new_x"]).max()
    errs = dict(x=np.abs(r["new_x"] - o["new_x"])[:, k].max(initial=0) / scale, u=np.abs(r["new_u"] - o["new_u"])[:, k].max(initial=0),
                cost=(np.abs(r["costs"] - o["costs"]) / (1 + np.abs(o["costs"])))[k].max(initial=0),
                du=(np.abs(r["full_du_norm"] - o["full_du_norm"]) / (1 + o["full_du_norm"]))[k].max(initial=0))
    if only:
        print("case %d kernel %s vouched %s T %d B %d mode %s: errs %s, |x| max %.3g, cost %s old %s" % (case, ekw["kernel"], ekw["nominal_on_dynamics"], T, B, mode, {k2: float("%.3g" % v) for k2, v in errs.items()}, np.abs(o["new_x"]).max(), np.round(o["costs"], 1), np.round(o["old_costs"], 1)))
    # (the 12/4 and 32/8 kernels price by the identity J_nominal + w0 + ...: the reported cost carries ~1e-7 |J_nominal| / |J|)
    ctol = 2e-4 + (3e-7 * float((np.abs(o["old_costs"]) / (1 + np.abs(o["costs"]))).max()) if ekw["kernel"].startswith(("dpp16", "mfma40")) else 0.0)
    # (new_u = u + k + K dx in float32: the error grows with how far the step moves the states, 1e-6 of it)
    move = float(np.abs(np.asarray(kw["cur_x"]) - o["new_x"]).max())
    xtol = 1e-3 + 3e-6 * move
    if f64:
        xtol, ctol = 1e-8 + 1e-12 * move, 1e-9
    viol = (flip & ~tie).any() or errs["x"] * scale > xtol * scale or errs["u"] > xtol or errs["cost"] > ctol or errs["du"] > 1e-3 or not np.isfinite(r["new_x"]).all()
    if viol:
        bad += 1
        print("VIOLATION case %d seed0 %d kernel %s ns %d nc %d T %d B %d mode %s opts %s ls (%g, %d): flips %s errs %s status %s" % (
            case, seed0, kernel, ns, nc, T, B, mode, {k2: (v if isinstance(v, (bool, int, float)) else "array") for k2, v in ekw.items()},
            kw["linesearch_decay"], kw["max_linesearch_iter"], np.nonzero(flip)[0].tolist(), {k2: float("%.3g" % v) for k2, v in errs.items()}, r["status"].tolist()))
print("cases %d violations %d  (%.0f s)" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
