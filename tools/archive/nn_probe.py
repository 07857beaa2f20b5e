import sys, time, torch
sys.path.insert(0, "mpc.pytorch_amd"); sys.path.insert(0, ".")
from mpc import _native
from mpc.dynamics import NNDynamics
from mpc._native import StepOptions
import bench
dev = "cuda:0"
be = _native.backend()
torch.manual_seed(0)
dyn = NNDynamics(12, 4, [100]).to(dev)
net = dyn.native_net(torch.empty(1, device=dev))
p = bench.make_problem(12, 4, 50, 4096, torch.float32, dev, seed=5, u_scale=0.3, clamp=1.0)
for name, fn in (("traj", lambda: be.mlp_traj_cost(p["x_init"], p["cur_u"], net)),
                 ("traj+cost", lambda: be.mlp_traj_cost(p["x_init"], p["cur_u"], net, C=p["C"], c=p["c"]))):
    w, ms, _ = bench.timed(fn, 30, 10)
    print(name, round(ms, 4))
from mpc._native import StepOptions
xs, _ = be.mlp_traj_cost(p["x_init"], p["cur_u"], net)
Fl, fl = be.mlp_linearize(net, xs[:-1].reshape(-1, 12), p["cur_u"][:-1].reshape(-1, 4))
F, f = Fl.view(49, 4096, 12, 16), fl.view(49, 4096, 12)
sw = be.lqr_step(p["x_init"], p["C"], p["c"], F, f, xs, p["cur_u"], StepOptions(u_lower=-1.0, u_upper=1.0, max_linesearch_iter=1), want_gains=True)
w, ms, rr = bench.timed(lambda: be.mlp_rollout(p["x_init"], p["C"], p["c"], sw["K"], sw["k"], xs, p["cur_u"], sw["old_costs"], StepOptions(u_lower=-1.0, u_upper=1.0), net), 30, 10)
print("search", round(ms, 4), float(rr["alphas"].mean()))
w, ms, _ = bench.timed(lambda: be.mlp_linearize(net, xs[:-1].reshape(-1, 12), p["cur_u"][:-1].reshape(-1, 4)), 30, 10)
print("linearize", round(ms, 4))
