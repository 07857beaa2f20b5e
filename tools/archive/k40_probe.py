#!/usr/bin/env python3
"""Why is config 5's nested KKT step slower than the plain step?  Times mpc_lqr_step at 32/8 T=64 B=1024 on variations of its
inputs (plain problem / the KKT backward's nested problem: zero nominal, c = -r, no f), alone and interleaved with a
memory-heavy kernel."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from mpc import _native
from mpc._native import StepOptions
be = _native.HipBackend()
NS, NC, T, B = 32, 8, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
p = bench.make_problem(NS, NC, T, B, torch.float32, "cuda:0", seed=5, u_scale=0.0)
o = StepOptions(nominal_on_dynamics=True, c_symmetric=True)
zx, zu, z0 = torch.zeros_like(p["cur_x"]), torch.zeros_like(p["cur_u"]), torch.zeros_like(p["x_init"])
negr = torch.randn_like(p["c"])
big = torch.empty(200 * 1024 * 1024, device="cuda:0")
res = {}
def t(name, fn, n=30):
    _, ms, _ = bench.timed(fn, n, 8)
    res[name] = round(ms * 1e3, 1)
t("plain", lambda: be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], o))
t("plain_nof_zero_x0", lambda: be.lqr_step(z0, p["C"], p["c"], p["F"], None, zx, zu, o))
t("nested(c=randn,zero nominal,no f)", lambda: be.lqr_step(z0, p["C"], negr, p["F"], None, zx, zu, o))
t("nested c scaled 1e-3", lambda: be.lqr_step(z0, p["C"], 1e-3 * negr, p["F"], None, zx, zu, o))
t("plain nominal, c=randn", lambda: be.lqr_step(p["x_init"], p["C"], negr, p["F"], p["f"], p["cur_x"], p["cur_u"], o))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
tot = 0.0
for i in range(38):
    big.fill_(1.0)
    ev[0].record()
    be.lqr_step(z0, p["C"], negr, p["F"], None, zx, zu, o)
    ev[1].record()
    torch.cuda.synchronize()
    if i >= 8: tot += ev[0].elapsed_time(ev[1])
res["nested, each behind an 800 MB fill"] = round(tot / 30 * 1e3, 1)
t("plain again", lambda: be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], o))
r = be.lqr_step(z0, p["C"], negr, p["F"], None, zx, zu, o)
res["nested alphas"] = [float(r["alphas"].min()), float(r["alphas"].max())]
print(json.dumps(res))
