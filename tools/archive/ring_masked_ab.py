import sys, os, torch
sys.path.insert(0, "mpc.pytorch_amd"); sys.path.insert(0, ".")
import bench
from mpc import _native
from mpc._native import StepOptions
be = _native.backend()
for B in (2048, 4096, 8192):
    p = bench.make_problem(12, 4, 50, B, torch.float32, "cuda:0", seed=5, u_scale=0.3, clamp=1.0)
    g = torch.Generator().manual_seed(1)
    mask = (torch.rand(50, B, 4, generator=g) < 0.3).cuda()
    for ring in ("2", "4"):
        os.environ["MPC_DPP16_RING"] = ring
        plan = be.plan_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], StepOptions(u_zero_I=mask, nominal_on_dynamics=True))
        w, ms, r = bench.timed(plan, 60, 30)
        print("masked step B", B, "ring", ring, "ms", round(ms, 4))
        r = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], StepOptions(u_lower=-1.0, u_upper=1.0, nominal_on_dynamics=True))
        gx, gu = torch.randn_like(r["new_x"]), torch.randn_like(r["new_u"])
        nx, nu = r["new_x"].clone(), r["new_u"].clone()
        w, ms, gg = bench.timed(lambda: be.kkt_backward(p["C"], p["c"], p["F"], p["f"], nx, nu, gx, gu, StepOptions(u_lower=-1.0, u_upper=1.0)), 30, 10)
        print("kkt_backward_bounded B", B, "ring", ring, "ms", round(ms, 4))
