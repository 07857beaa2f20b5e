#!/usr/bin/env python3
"""Secondary timings at the headline shape (not the bench.py contract): the KKT backward
(LQRStepFn.backward), a full MPC.forward solve, and the small-shape configs."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import bench
from mpc import _native, mpc
from mpc._native import StepOptions
from mpc.mpc import QuadCost, LinDx

def timed(fn, n=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]

def main():
    be = _native.HipBackend()
    out = {}
    for bounded in (False, True):
        p = bench.make_problem(12, 4, 50, 4096, torch.float32, "cuda:0", seed=5, u_scale=0.3 if bounded else 0.0,
                               clamp=1.0 if bounded else None)
        opts = StepOptions(u_lower=-1.0, u_upper=1.0) if bounded else StepOptions()
        r = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], opts)
        gx, gu = torch.randn_like(r["new_x"]), torch.randn_like(r["new_u"])
        key = "bounded" if bounded else "unbounded"
        out["kkt_backward_ms_" + key] = timed(lambda: be.kkt_backward(p["C"], p["c"], p["F"], p["f"], r["new_x"], r["new_u"], gx, gu, opts))
        out["lqr_step_ms_" + key] = timed(lambda: be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], opts))
        ctrl = mpc.MPC(12, 4, 50, u_lower=-1.0 if bounded else None, u_upper=1.0 if bounded else None, lqr_iter=5,
                       verbose=-1, exit_unconverged=False, detach_unconverged=False)
        out["mpc_forward_5iter_ms_" + key] = timed(lambda: ctrl(p["x_init"], QuadCost(p["C"], p["c"]), LinDx(p["F"], p["f"])), n=5, warm=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
