#!/usr/bin/env python3
"""Randomised parity stress of the fused KKT backward at n_state = 32, n_ctrl = 8 (mpc_lqr_kkt_fused: lqr_kkt_fused_mfma40_kernel
+ kkt_outer_kernel) against the float64 oracle: random horizons, batches, bounds (none / scalar / tensor), with and without f,
solutions a few LQR steps from a random nominal.  One line per case; exits non-zero on a violation.
usage: python tools/stress_kkt40.py [cases]"""
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
rng = np.random.default_rng(2026)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
h64 = lambda t: None if t is None else t.detach().cpu().numpy().astype(np.float64)
for case in range(cases):
    T = int(rng.choice([1, 2, 3, 4, 5, 9, 17, 33, 64, 70]))
    B = int(rng.choice([1, 2, 7, 33, 64, 130]))
    kind = ("none", "scalar", "tensor")[case % 3]
    with_f = bool(rng.integers(0, 2)) and T > 1
    ub = float(rng.choice([0.3, 0.5, 1.0]))
    p = bench.make_problem(32, 8, max(T, 2), B, torch.float32, "cuda:0", seed=1000 + case, u_scale=0.3, clamp=ub if kind != "none" else None)
    if T == 1:
        p = {k: (v[:1].contiguous() if k in ("C", "c", "cur_x", "cur_u") else (v[:0].contiguous() if k in ("F", "f") else v)) for k, v in p.items()}
    f = p["f"] if with_f else None
    if not with_f and T > 1:
        from mpc import util
        from mpc.mpc import LinDx
        p["cur_x"] = util.get_traj(T, p["cur_u"], p["x_init"], LinDx(p["F"], None))
    kw, lo, hi = {}, None, None
    if kind == "scalar":
        # (the oracle gets the bound as float32 holds it: a control ON the bound is on it to 1e-8 only then -- 0.3f - 0.3 = 1.2e-8)
        kw, lo, hi = dict(u_lower=-ub, u_upper=ub), float(np.float32(-ub)), float(np.float32(ub))
    elif kind == "tensor":
        g = torch.Generator().manual_seed(case)
        lo_t = (-ub - 0.2 * torch.rand(T, B, 8, generator=g)).cuda(); hi_t = (ub + 0.2 * torch.rand(T, B, 8, generator=g)).cuda()
        kw, lo, hi = dict(u_lower=lo_t, u_upper=hi_t), h64(lo_t), h64(hi_t)
        p["cur_u"] = torch.maximum(torch.minimum(p["cur_u"], hi_t), lo_t)
        if T > 1:
            from mpc import util
            from mpc.mpc import LinDx
            p["cur_x"] = util.get_traj(T, p["cur_u"], p["x_init"], LinDx(p["F"], f))
    x, u = p["cur_x"], p["cur_u"]
    for _ in range(3):
        r = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"] if T > 1 else None, f, x, u, StepOptions(**kw))
        x, u = r["new_x"], r["new_u"]
    gx, gu = torch.randn_like(x), torch.randn_like(u)
    got = be.kkt_backward(p["C"], p["c"], p["F"] if T > 1 else torch.empty(0, B, 32, 40, device="cuda:0"), f, x, u, gx, gu,
                          StepOptions(c_symmetric=True, **kw))
    torch.cuda.synchronize()
    o = O.kkt_backward(h64(p["C"]), h64(p["c"]), h64(p["F"]) if T > 1 else np.zeros((0, B, 32, 40)), h64(f), h64(x), h64(u), h64(gx), h64(gu), lo, hi, lockstep=False)
    worst = {}
    for k in ("dx_init", "dC", "dc", "dF", "df"):
        if o[k] is None or o[k].size == 0:
            continue
        a = got[k].cpu().numpy().astype(np.float64)
        ax = tuple(i for i in range(a.ndim) if i != (0 if k == "dx_init" else 1))
        scale = np.maximum(1.0, np.abs(o[k]).max(axis=ax, keepdims=True))
        worst[k] = float((np.abs(a - o[k]) / scale).max()) if np.isfinite(a).all() else float("inf")
    act = float(((np.abs(h64(u) - lo) <= 1e-8) | (np.abs(h64(u) - hi) <= 1e-8)).mean()) if kind != "none" else 0.0
    over = [k for k, v in worst.items() if not v < 5e-4]
    bad += len(over)
    print(json.dumps({"case": case, "T": T, "B": B, "bounds": kind, "f": with_f, "active_share": round(act, 3), "worst_rel": {k: float("%.2e" % v) for k, v in worst.items()}, "over_tol": over}), flush=True)
print("violations:", bad)
sys.exit(1 if bad else 0)
