#!/usr/bin/env python3
"""One row per BASELINE.json configuration on one MI355X (not the bench.py contract: that is config 4).
Writes gpurun_out/bench_configs.json; copy to profiles/ to keep."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from mpc import _native, mpc
from mpc._native import StepOptions
from mpc.mpc import QuadCost, LinDx
from tools.bench_extra import timed
from tools.bench_ilqr_env import run as run_env

be = _native.HipBackend()
rows = []


def step_row(name, ns, nc, T, B, opts, **kw):
    p = bench.make_problem(ns, nc, T, B, torch.float32, "cuda:0", seed=11, **kw)
    ms = timed(lambda: be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], opts), n=10)
    r = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], opts)
    gx, gu = torch.randn_like(r["new_x"]), torch.randn_like(r["new_u"])
    kkt = timed(lambda: be.kkt_backward(p["C"], p["c"], p["F"], p["f"], r["new_x"], r["new_u"], gx, gu, opts), n=5)
    ab = bench.algorithmic_bytes_per_problem(ns, nc, T) * B
    rows.append({"config": name, "lqr_step_us": round(ms * 1e3, 1), "kkt_backward_us": round(kkt * 1e3, 1),
                 "problem_steps_per_s": round(B * T / (ms * 1e-3)), "algorithmic_GBps": round(ab / (ms * 1e-3) / 1e9, 1),
                 "frac_of_8TBps": round(ab / (ms * 1e-3) / 8e12, 4)})


g = torch.Generator().manual_seed(0)
lo = -torch.rand(10, 8, 2, generator=g).cuda(); hi = torch.rand(10, 8, 2, generator=g).cuda()
step_row("1: time-varying LQR ns=4 nc=2 T=10 B=8, tensor bounds (latency-bound: 21 KB of data)", 4, 2, 10, 8, StepOptions(u_lower=lo, u_upper=hi))
for r in (run_env("pendulum", 1024, 20, 10, 10), run_env("cartpole", 4096, 25, 10, 10)):
    rows.append({"config": ("2: pendulum iLQR ns=3 nc=1 T=20 B=1024" if r["config"] == "pendulum" else "3: cart-pole iLQR ns=5 nc=1 T=25 B=4096, box constraints"),
                 "mpc_forward_10_iterations_ms": r["kernel_path"]["ms_per_solve"], "per_ilqr_iteration_us": round(1e3 * r["kernel_path"]["ms_per_ilqr_iteration"], 1),
                 "problem_steps_per_s": r["kernel_path"]["problem_steps_per_s"],
                 "host_driven_module_path_ms": r["module_path"]["ms_per_solve"], "speedup": r["speedup"]})
step_row("4a: ns=12 nc=4 T=50 B=4096 unbounded (the bench.py line)", 12, 4, 50, 4096, StepOptions())
step_row("4b: ns=12 nc=4 T=50 B=4096 bounds +-1", 12, 4, 50, 4096, StepOptions(u_lower=-1.0, u_upper=1.0), u_scale=0.3, clamp=1.0)
step_row("5: ns=32 nc=8 T=64 B=1024 (= 8192 over 8 GPUs), MFMA tile path", 32, 8, 64, 1024, StepOptions())
step_row("5 (one GPU takes all 8192)", 32, 8, 64, 8192, StepOptions())
for r in rows:
    print(json.dumps(r))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "bench_configs.json"), "w"), indent=1)
