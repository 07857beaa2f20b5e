"""KKT backward (mpc/lqr_step.py:312-407) on shapes between 12/4 and 32/8 beside exactly 32/8, B=1024 T=64
    python tools/pad40_kkt_bench.py            (on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd"))
import torch
import bench
from mpc._native import HipBackend, StepOptions
dev = torch.device("cuda", 0)
be = HipBackend()
T, B = 64, 1024
for ns, nc in ((32, 8), (13, 4), (16, 4), (20, 5), (24, 8)):
    for bounded in (False, True):
        p = bench.make_problem(ns, nc, T, B, torch.float32, dev, seed=60 + ns, u_scale=0.3 if bounded else 0.0, clamp=1.0 if bounded else None)
        o = (StepOptions(u_lower=-1.0, u_upper=1.0, nominal_on_dynamics=True, c_symmetric=True) if bounded
             else StepOptions(nominal_on_dynamics=True, c_symmetric=True))
        r = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], o)
        gx, gu = torch.randn_like(r["new_x"]), torch.randn_like(r["new_u"])
        nx, nu = r["new_x"].clone(), r["new_u"].clone()
        kfn = be.plan_kkt_backward(p["C"], p["c"], p["F"], p["f"], nx, nu, gx, gu, o)
        fused = kfn is not None
        if kfn is None:
            kfn = lambda: be.kkt_backward(p["C"], p["c"], p["F"], p["f"], nx, nu, gx, gu, o)
        wall, ms, ms_all, g = bench.timed_sustained(kfn, 30, 40)
        kb = bench.kkt_algorithmic_bytes_per_problem(ns, nc, T) * B
        print("%2d/%d T=%d B=%d %s KKT backward (%s): %.1f us  frac %.3f" % (ns, nc, T, B, "bounded" if bounded else "unbounded", "fused" if fused else "three launches",
                                                                          ms * 1e3, kb / (ms * 1e-3) / 8e12), flush=True)
