#!/usr/bin/env python3
"""Config 5 of BASELINE.json (n_state=32, n_ctrl=8, T=64) through the generic kernels: forward + KKT backward."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import bench
from mpc import _native
from mpc._native import StepOptions
from tools.bench_extra import timed
be = _native.HipBackend()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
p = bench.make_problem(32, 8, 64, B, torch.float32, "cuda:0", seed=9)
opts = StepOptions()
r = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], opts)
gx, gu = torch.randn_like(r["new_x"]), torch.randn_like(r["new_u"])
out = {"B": B,
       "lqr_step_ms": timed(lambda: be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], opts), n=5, warm=1),
       "kkt_backward_ms": timed(lambda: be.kkt_backward(p["C"], p["c"], p["F"], p["f"], r["new_x"], r["new_u"], gx, gu, opts), n=5, warm=1)}
out["sweep_only_ms"] = timed(lambda: be.lqr_sweep(p["x_init"], p["C"], p["c"], p["F"], p["cur_x"], p["cur_u"], opts), n=5, warm=1)
out["sweep_plus_rollout_split_ms"] = timed(lambda: be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], opts,
                                                               rollout_problem=(p["C"], p["c"], p["F"], p["f"])), n=5, warm=1)
bo = StepOptions(u_lower=-1.0, u_upper=1.0)
out["lqr_step_bounded_ms"] = timed(lambda: be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], bo), n=5, warm=1)
out["sweep_only_bounded_ms"] = timed(lambda: be.lqr_sweep(p["x_init"], p["C"], p["c"], p["F"], p["cur_x"], p["cur_u"], bo), n=5, warm=1)
out["problem_steps_per_s"] = B * 64 / (out["lqr_step_ms"] * 1e-3)
out["algorithmic_GBps"] = bench.algorithmic_bytes_per_problem(32, 8, 64) * B / (out["lqr_step_ms"] * 1e-3) / 1e9
print(json.dumps(out))
