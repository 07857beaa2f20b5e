#!/usr/bin/env python3
"""Headline shape (12/4, T=50, B=4096): the step and the fused backward with warm address translations (back-to-back launches,
what bench.py times) and cold ones (a 16 us kernel touching one byte in every 4 KiB page of 800 MB in front of every launch),
on both sweep rings of the 12/4 kernel (MPC_DPP16_RING, needs MPC_DPP16_RING_DYNAMIC=1)."""
import json, os, sys
os.environ["MPC_DPP16_RING_DYNAMIC"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import torch, bench
from mpc import _native
from mpc._native import StepOptions
be = _native.HipBackend()
NS, NC, T, B = 12, 4, 50, (int(sys.argv[1]) if len(sys.argv) > 1 else 4096)
big = torch.zeros(800 * 1024 * 1024, dtype=torch.uint8, device="cuda:0")
view = big[::4096]
_, t0, _ = bench.timed(lambda: view.sum(), 50, 10)
res = {"toucher_us": round(t0 * 1e3, 1)}
for bounded in (False, True):
    p = bench.make_problem(NS, NC, T, B, torch.float32, "cuda:0", seed=5, u_scale=0.3 if bounded else 0.0, clamp=1.0 if bounded else None)
    a = (p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"])
    kw = dict(u_lower=-1.0, u_upper=1.0) if bounded else {}
    o = StepOptions(nominal_on_dynamics=True, c_symmetric=True, **kw)
    plan = be.plan_step(*a, o) if hasattr(be, "plan_step") else (lambda: be.lqr_step(*a, o))
    r = be.lqr_step(*a, o)
    gx, gu = torch.randn_like(r["new_x"]), torch.randn_like(r["new_u"])
    nx, nu = r["new_x"].clone(), r["new_u"].clone()
    key = "bounded" if bounded else "unbounded"
    for ring in ("2", "4"):
        os.environ["MPC_DPP16_RING"] = ring
        _, w, _ = bench.timed(plan, 100, 100)
        _, c, _ = bench.timed(lambda: (view.sum(), plan())[1], 100, 20)
        res["step_%s_ring%s" % (key, ring)] = {"warm_us": round(w * 1e3, 1), "cold_us": round((c - t0) * 1e3, 1)}
    os.environ.pop("MPC_DPP16_RING")
    kk = lambda: be.kkt_backward(p["C"], p["c"], p["F"], p["f"], nx, nu, gx, gu, o)
    _, w, _ = bench.timed(kk, 50, 20)
    _, c, _ = bench.timed(lambda: (view.sum(), kk())[1], 50, 20)
    res["kkt_fused_%s" % key] = {"warm_us": round(w * 1e3, 1), "cold_us": round((c - t0) * 1e3, 1)}
print(json.dumps(res))
