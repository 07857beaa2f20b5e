import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import torch, bench
from mpc import _native
be = _native.HipBackend()
for (ns, nc, T, B) in ((32, 8, 64, 1024), (32, 8, 64, 8192), (20, 4, 50, 4096)):
    p = bench.make_problem(ns, nc, T, B, torch.float32, "cuda:0", seed=9)
    _, ms, _ = bench.timed(lambda: be.traj_cost(p["x_init"], p["cur_u"], p["F"], p["f"]), 30, 8)
    fb = (T - 1) * B * ns * (ns + nc) * 4
    print("get_traj", ns, nc, T, B, "us", round(ms * 1e3, 1), "TB/s of F", round(fb / (ms * 1e-3) / 1e12, 2))
