"""One MPC.forward at the headline shape under `rocprofv3 --kernel-trace --output-format csv`: which kernels, in what order,
with what gaps.   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -o tr -- python tools/trace_mpc_forward.py [bounded | cfg5 | pendulum | cartpole | nn]
then   python tools/trace_mpc_forward.py --read gpurun_out/tr"""
import sys, os, glob, csv, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)

if "--read" in sys.argv:
    f = glob.glob(os.path.join(sys.argv[-1], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # the last call: walk back from the end to the last trajectory kernel
    starts = [i for i, r in enumerate(rows) if "traj_" in r["Kernel_Name"] or "nn_rollout" in r["Kernel_Name"] and "traj" in r["Kernel_Name"].lower()]
    if not starts:          # (the network solves: the trajectory kernel is the first launch of a solve -- find the last select, walk back one solve)
        sel = [i for i, r in enumerate(rows) if "select_best" in r["Kernel_Name"]]
        starts = [sel[-6] + 1 if len(sel) > 5 else 0]
    rows = rows[starts[-1]:]
    t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        nm = re.sub(r"\(anonymous namespace\)::|mpclqr::|void |at::native::", "", r["Kernel_Name"])
        print("%8.1f us  +gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, nm[:90]))
        prev_end = e
    print("total %.1f us" % ((prev_end - t0) / 1e3))
    sys.exit(0)

import torch, bench
from mpc import mpc
from mpc.mpc import QuadCost, LinDx
bounded = "bounded" in sys.argv
env = [k for k in ("pendulum", "cartpole") if k in sys.argv]
if env:                                            # BASELINE configs 2 / 3: the shipped simulators, 10 iterations
    from tools.bench_ilqr_env import problem as env_problem
    B, T = (1024, 20) if env[0] == "pendulum" else (4096, 25)
    dxm, _plain, x0, Q, pp = env_problem(env[0], B, T)
    ctrl = mpc.MPC(dxm.n_state, 1, T, u_lower=dxm.lower, u_upper=dxm.upper, lqr_iter=10, verbose=-1, exit_unconverged=False,
                   detach_unconverged=False, linesearch_decay=dxm.linesearch_decay, max_linesearch_iter=dxm.max_linesearch_iter,
                   grad_method=mpc.GradMethods.AUTO_DIFF, eps=1e-12, backprop=False, not_improved_lim=10 ** 6)
    for _ in range(4):
        ctrl(x0, QuadCost(Q, pp), dxm)
    torch.cuda.synchronize()
    sys.exit(0)
# (round 5: bench.py's own problems -- seed 5, the box-constrained one with its nominal u ~ 0.3 N clamped -- and 30 solves, so
# that the LAST one, which --read prints, runs in the sustained state like the bench rows)
if "nn" in sys.argv:                               # bench.py's nn_mpc_forward_5iter: NNDynamics(12, 4, [100]) as the dynamics
    from mpc.dynamics import NNDynamics
    torch.manual_seed(0)
    dyn = NNDynamics(12, 4, [100], activation="sigmoid").to("cuda:0")
    p = bench.make_problem(12, 4, 50, 4096, torch.float32, "cuda:0", seed=5, u_scale=0.3, clamp=1.0)
    ctrl = mpc.MPC(12, 4, 50, u_lower=-1.0, u_upper=1.0, lqr_iter=5, verbose=-1, exit_unconverged=False, detach_unconverged=False,
                   grad_method=mpc.GradMethods.ANALYTIC, backprop=False, u_init=p["cur_u"].clone())
    with torch.no_grad():
        for _ in range(12):
            ctrl(p["x_init"], QuadCost(p["C"], p["c"]), dyn)
    torch.cuda.synchronize()
    sys.exit(0)
ns_, nc_, T_, B_ = (32, 8, 64, 1024) if "cfg5" in sys.argv else (12, 4, 50, 4096)          # (cfg5: BASELINE config 5 per GPU)
p = bench.make_problem(ns_, nc_, T_, B_, torch.float32, "cuda:0", seed=9 if "cfg5" in sys.argv else 5, u_scale=0.3 if bounded else 0.0, clamp=1.0 if bounded else None)
ctrl = mpc.MPC(ns_, nc_, T_, u_lower=-1.0 if bounded else None, u_upper=1.0 if bounded else None, lqr_iter=5, verbose=-1,
               exit_unconverged=False, detach_unconverged=False, backprop=False)
cost, dx = QuadCost(p["C"], p["c"]), LinDx(p["F"], p["f"])
for _ in range(30):
    ctrl(p["x_init"], cost, dx)
torch.cuda.synchronize()
