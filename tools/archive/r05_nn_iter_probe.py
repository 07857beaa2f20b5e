#!/usr/bin/env python3
"""Round 5: the line-search depths of the iterations of bench.py's network solve (NNDynamics(12, 4, [100]), B = 4096, T = 50), and how
many problems of one 16-problem wavefront keep searching after two failed trials."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import bench
from mpc import _native, util
from mpc._native import StepOptions
from mpc.dynamics import NNDynamics
be = _native.HipBackend()
torch.manual_seed(0)
dyn = NNDynamics(12, 4, [100], activation="sigmoid").to("cuda:0")
p = bench.make_problem(12, 4, 50, 4096, torch.float32, "cuda:0", seed=5, u_scale=0.3, clamp=1.0)
net = dyn.native_net(p["x_init"])
ua = p["cur_u"].clone()
xa = util.get_traj(50, ua, p["x_init"], dyn).contiguous()
xb, ub = torch.empty_like(xa), torch.empty_like(ua)
run, outs, vouch = be.plan_network_iteration(p["x_init"], p["C"], p["c"], net, StepOptions(u_lower=-1.0, u_upper=1.0), ((xa, ua), (xb, ub)))
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); r = run(i % 2); b_.record(); torch.cuda.synchronize()
    al = r["alphas"].cpu().numpy()
    depth = np.rint(np.log(al) / np.log(0.2)).astype(int)
    dc = (r["costs"] - r["old_costs"]).cpu().numpy()
    deep = (depth >= 2).reshape(-1, 16).sum(1)
    print("iteration %d: %.0f us  depth counts %s  worse at the end: %d  waves by problems deeper than 2 trials: %s  mean cost %.6g" % (
        i, a.elapsed_time(b_) * 1e3, np.bincount(depth, minlength=10).tolist(), int((dc > 0).sum()), np.bincount(deep).tolist(), float(r["costs"].mean())))
