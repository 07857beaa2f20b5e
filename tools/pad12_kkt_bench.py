"""KKT backward (mpc/lqr_step.py:255-368) on shapes up to 12/4 beside exactly 12/4: what the padded shapes' backward costs
    python tools/pad12_kkt_bench.py            (on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd"))
import torch
import bench
from mpc._native import HipBackend, StepOptions
dev = torch.device("cuda", 0)
be = HipBackend()
T, B = 64, 4096
for ns, nc in ((12, 4), (10, 3), (8, 4), (12, 2)):
    for bounded in (False, True):
        p = bench.make_problem(ns, nc, T, B, torch.float32, dev, seed=60 + ns, u_scale=0.3 if bounded else 0.0, clamp=1.0 if bounded else None)
        o = (StepOptions(u_lower=-1.0, u_upper=1.0, nominal_on_dynamics=True, c_symmetric=True) if bounded
             else StepOptions(nominal_on_dynamics=True, c_symmetric=True))
        r = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], o)
        gx, gu = torch.randn_like(r["new_x"]), torch.randn_like(r["new_u"])
        nx, nu = r["new_x"].clone(), r["new_u"].clone()
        kfn = be.plan_kkt_backward(p["C"], p["c"], p["F"], p["f"], nx, nu, gx, gu, o)
        if kfn is None:
            kfn = lambda: be.kkt_backward(p["C"], p["c"], p["F"], p["f"], nx, nu, gx, gu, o)
        wall, ms, ms_all, g = bench.timed_sustained(kfn, 40, 60)
        kb = bench.kkt_algorithmic_bytes_per_problem(ns, nc, T) * B
        print("%2d/%d T=%d B=%d %s KKT backward: %.1f us  frac %.3f" % (ns, nc, T, B, "bounded" if bounded else "unbounded", ms * 1e3, kb / (ms * 1e-3) / 8e12), flush=True)
