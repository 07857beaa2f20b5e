#!/usr/bin/env python3
"""`_module_rollout` with an arbitrary nn.Module as dynamics: graph replay against the eager loop (GPU box)."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mpc import _native, lqr_step, mpc
from mpc._native import StepOptions
import bench
from test_gpu_nn import _TanhDynamics
dev = "cuda:0"
ns, nc, T, B = 12, 4, 50, 4096
be = _native.backend()
p = bench.make_problem(ns, nc, T, B, torch.float32, dev, seed=5, u_scale=0.3, clamp=1.0)
dyn = _TanhDynamics(ns, nc).to(dev)
opts = StepOptions(u_lower=-1.0, u_upper=1.0)
r = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], StepOptions(u_lower=-1.0, u_upper=1.0, max_linesearch_iter=1), want_gains=True)
cost = mpc.QuadCost(p["C"], p["c"])
def run():
    return lqr_step._module_rollout(ns, nc, T, p["x_init"], r["K"], r["k"], p["cur_x"], p["cur_u"], r["old_costs"], cost, dyn, opts)
def timed(n=5):
    run(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): run()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
res = {"graph_ms": timed()}
a = run()
os.environ["MPC_NO_ROLLOUT_GRAPH"] = "1"
res["eager_ms"] = timed()
b = run()
res["passes_alpha_mean"] = float(a[5].mean())
res["max_abs_diff_u"] = float((a[1] - b[1]).abs().max())
print(json.dumps(res))
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "module_rollout_bench.json"), "w"))
