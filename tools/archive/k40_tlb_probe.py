#!/usr/bin/env python3
"""Is it the TLB?  The 32/8 fused backward WITHOUT its outer-product kernel (MPC_LQR_HIP_LIB=variants/lib_k40_noouter.so),
alone and behind a kernel that touches one byte in every 4 KiB page of an 800 MB / 3.2 GB buffer (no bandwidth, no power)."""
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
for mb in (0, 800, 3200, 0):
    big = torch.zeros(max(mb, 1) * 1024 * 1024, dtype=torch.uint8, device="cuda:0")
    view = big[::4096]
    def touch():
        if mb:
            view.sum()
    def both():
        touch()
        return be.kkt_backward(p["C"], p["c"], p["F"], p["f"], nx, nu, gx, gu, o)
    _, ms, _ = bench.timed(both, 30, 8)
    _, ms0, _ = bench.timed(touch, 30, 8) if mb else (0, 0.0, 0)
    res.setdefault("touch_%d_MB" % mb, []).append({"both_us": round(ms * 1e3, 1), "touch_us": round(ms0 * 1e3, 1), "backward_us": round((ms - ms0) * 1e3, 1)})
    del big, view
print(json.dumps(res))
