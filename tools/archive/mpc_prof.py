import os, sys, time, torch
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import bench
from mpc import mpc
from mpc.mpc import QuadCost, LinDx
p = bench.make_problem(12, 4, 50, 4096, torch.float32, "cuda:0", seed=5)
def run(iters):
    ctrl = mpc.MPC(12, 4, 50, lqr_iter=iters, verbose=-1, exit_unconverged=False, detach_unconverged=False)
    for _ in range(2): ctrl(p["x_init"], QuadCost(p["C"], p["c"]), LinDx(p["F"], p["f"]))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ctrl(p["x_init"], QuadCost(p["C"], p["c"]), LinDx(p["F"], p["f"]))
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 5 * 1e3
for it in (1, 2, 5, 10, 20):
    print("lqr_iter", it, "ms", round(run(it), 3))
import cProfile, pstats
ctrl = mpc.MPC(12, 4, 50, lqr_iter=10, verbose=-1, exit_unconverged=False, detach_unconverged=False)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): ctrl(p["x_init"], QuadCost(p["C"], p["c"]), LinDx(p["F"], p["f"]))
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
