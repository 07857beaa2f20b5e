import os, sys, torch, numpy as np
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import bench
from mpc import _native
from mpc._native import StepOptions
be = _native.HipBackend()
for seed in (5, 1000):
    p = bench.make_problem(12, 4, 50, 4096, torch.float32, "cuda:0", seed=seed, u_scale=0.3, clamp=1.0)
    r = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], StepOptions(u_lower=-1.0, u_upper=1.0))
    a = r["alphas"].cpu().numpy()
    vals, cnt = np.unique(np.round(a, 6), return_counts=True)
    print("seed", seed, dict(zip(vals.tolist(), cnt.tolist())), "waves with any reject:", int((a.reshape(-1, 4) < 1).any(1).sum()), "of 1024")
