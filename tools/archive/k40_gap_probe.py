#!/usr/bin/env python3
"""Why does the 32/8 fused backward take 150 us longer right behind its own outer-product kernel?  Times the pair with an idle
gap (a one-block spin kernel: no memory traffic, no power) of various lengths behind the outer-product kernel."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import torch, bench
from mpc import _native
from mpc._native import StepOptions
be = _native.HipBackend()
B = 1024
p = bench.make_problem(32, 8, 64, B, torch.float32, "cuda:0", seed=9, u_scale=0.3, clamp=1.0)
o = StepOptions(nominal_on_dynamics=True, c_symmetric=True)
r = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], o)
gx, gu = torch.randn_like(r["new_x"]), torch.randn_like(r["new_u"])
nx, nu = r["new_x"].clone(), r["new_u"].clone()
res = {}
for cycles in (0, 100000, 400000, 1600000, 0):
    def pair():
        g = be.kkt_backward(p["C"], p["c"], p["F"], p["f"], nx, nu, gx, gu, o)
        if cycles:
            torch.cuda._sleep(cycles)
        return g
    def gap():
        if cycles:
            torch.cuda._sleep(cycles)
    _, ms, _ = bench.timed(pair, 30, 8)
    _, ms0, _ = bench.timed(gap, 30, 8) if cycles else (0, 0.0, 0)
    res.setdefault("gap_cycles_%d" % cycles, []).append({"pair_plus_gap_us": round(ms * 1e3, 1), "gap_us": round(ms0 * 1e3, 1), "pair_us": round((ms - ms0) * 1e3, 1)})
print(json.dumps(res))
