"""ONE shape's KKT backward at B=1024 T=64, a few launches: the command a rocprofv3 kernel trace splits by kernel
    python tools/pad40_kkt_one.py NS NC [bounded]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd"))
import torch
import bench
from mpc._native import HipBackend, StepOptions
ns, nc = int(sys.argv[1]), int(sys.argv[2])
bounded = len(sys.argv) > 3
dev = torch.device("cuda", 0)
be = HipBackend()
T, B = 64, 1024
p = bench.make_problem(ns, nc, T, B, torch.float32, dev, seed=60 + ns, u_scale=0.3 if bounded else 0.0, clamp=1.0 if bounded else None)
o = (StepOptions(u_lower=-1.0, u_upper=1.0, nominal_on_dynamics=True, c_symmetric=True) if bounded else StepOptions(nominal_on_dynamics=True, c_symmetric=True))
r = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], o)
gx, gu = torch.randn_like(r["new_x"]), torch.randn_like(r["new_u"])
nx, nu = r["new_x"].clone(), r["new_u"].clone()
kfn = be.plan_kkt_backward(p["C"], p["c"], p["F"], p["f"], nx, nu, gx, gu, o) or (lambda: be.kkt_backward(p["C"], p["c"], p["F"], p["f"], nx, nu, gx, gu, o))
for _ in range(60):
    kfn()
torch.cuda.synchronize()
