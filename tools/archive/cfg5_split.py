import sys, torch
sys.path.insert(0, "mpc.pytorch_amd"); sys.path.insert(0, ".")
from mpc import _native
from mpc._native import StepOptions
import bench
be = _native.backend()
for B in (1024, 8192):
    p = bench.make_problem(32, 8, 64, B, torch.float32, "cuda:0", seed=9, on_device=True)
    plan = be.plan_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], StepOptions(nominal_on_dynamics=True))
    w, ms, r = bench.timed(plan, 40, 20)
    print("B", B, "ms", round(ms, 4))
p = bench.make_problem(32, 8, 64, 1024, torch.float32, "cuda:0", seed=9, on_device=True)
plan = be.plan_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], StepOptions(u_lower=-1.0, u_upper=1.0))
w, ms, r = bench.timed(plan, 40, 20)
print("bounded B 1024 ms", round(ms, 4), "unconverged", int((r["status"] & 1).sum()))
