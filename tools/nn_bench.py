#!/usr/bin/env python3
"""Timings of the NNDynamics path on the GPU box: the three kernels of csrc/nn_dynamics.hip against this package's
host-driven path (the module called timestep by timestep; torch's batched grad_input), and a whole MPC.forward.
usage: python tools/nn_bench.py [B] [T] [hidden]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd"))
sys.path.insert(0, ROOT)


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    from mpc import _native, mpc, util
    from mpc.dynamics import NNDynamics
    from mpc._native import StepOptions
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    ns, nc = 12, 4
    n = ns + nc
    dev = "cuda:0"
    torch.manual_seed(0)
    dyn = NNDynamics(ns, nc, [H], activation="sigmoid").to(dev)
    be = _native.backend()
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(B, ns, generator=g).to(dev)
    u0 = (0.3 * torch.randn(T, B, nc, generator=g)).clamp(-1, 1).to(dev)
    A = torch.randn(T, B, n, n, generator=g).to(dev)
    C = A.transpose(2, 3).matmul(A).contiguous()
    c = torch.randn(T, B, n, generator=g).to(dev)
    net = dyn.native_net(x0)
    xs, _ = be.mlp_traj_cost(x0, u0, net)
    X, U = xs[:-1].reshape(-1, ns), u0[:-1].reshape(-1, nc)
    Fl, fl = be.mlp_linearize(net, X, U)
    F, f = Fl.view(T - 1, B, ns, n), fl.view(T - 1, B, ns)
    opts = StepOptions(u_lower=-1.0, u_upper=1.0)
    sw = be.lqr_step(x0, C, c, F, f, xs, u0, StepOptions(u_lower=-1.0, u_upper=1.0, max_linesearch_iter=1), want_gains=True)
    res = {"B": B, "T": T, "hidden": H}
    res["kernel_get_traj_ms"] = timed(lambda: be.mlp_traj_cost(x0, u0, net))
    res["kernel_linearize_ms"] = timed(lambda: be.mlp_linearize(net, X, U))
    res["kernel_sweep_ms"] = timed(lambda: be.lqr_step(x0, C, c, F, f, xs, u0, StepOptions(u_lower=-1.0, u_upper=1.0, max_linesearch_iter=1), want_gains=True))
    res["kernel_rollout_ms"] = timed(lambda: be.mlp_rollout(x0, C, c, sw["K"], sw["k"], xs, u0, sw["old_costs"], opts, net))

    # host-driven
    class Plain(torch.nn.Module):          # the same network without the kernel hook
        def __init__(self, d):
            super().__init__()
            self.d = d

        def forward(self, x, u):
            return self.d(x, u)

        def grad_input(self, x, u):
            return self.d.grad_input(x, u)
    plain = Plain(dyn)
    res["host_get_traj_ms"] = timed(lambda: util.get_traj(T, u0, x0, plain), n=3, warm=1)

    def torch_lin():
        with torch.no_grad():
            nx = dyn(X, U)
            R, S = dyn.grad_input(X, U)
            return torch.cat((R, S), 2), nx - util.bmv(R, X) - util.bmv(S, U)
    try:
        res["host_linearize_ms"] = timed(torch_lin, n=3, warm=1)
    except RuntimeError as e:            # [N, hidden, n] intermediates may not fit
        res["host_linearize_ms"] = None
        res["host_linearize_error"] = str(e)[:80]
    from mpc.lqr_step import _module_rollout
    res["host_rollout_ms"] = timed(lambda: _module_rollout(ns, nc, T, x0, sw["K"], sw["k"], xs, u0, sw["old_costs"],
                                                         mpc.QuadCost(C, c), plain, opts), n=3, warm=1)
    # whole solves, 5 iLQR iterations
    def solve(d):
        ctrl = mpc.MPC(ns, nc, T, u_lower=-1.0, u_upper=1.0, lqr_iter=5, verbose=-1, exit_unconverged=False,
                       detach_unconverged=False, grad_method=mpc.GradMethods.ANALYTIC, u_init=u0.clone(), backprop=False)
        with torch.no_grad():
            return ctrl(x0, mpc.QuadCost(C, c), d)
    res["mpc_forward_5iter_kernels_ms"] = timed(lambda: solve(dyn), n=3, warm=1)
    res["mpc_forward_5iter_host_ms"] = timed(lambda: solve(plain), n=2, warm=1)
    a, b = solve(dyn), solve(plain)
    res["solve_cost_rel_diff"] = float(((a[2] - b[2]).abs() / (1 + b[2].abs())).max())
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "nn_bench_B%d_T%d_H%d.json" % (B, T, H)), "w"), indent=1)


if __name__ == "__main__":
    main()
