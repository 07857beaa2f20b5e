"""Clock stamps of ONE timestep of the Riccati recursion (phase P2) of the row-per-problem kernel, wavefront 0: needs the library
built with -DMPC_W1_PROF (csrc/lqr_wave1_body.h), which parks the stamps in qp_iters[8..12]:
    MPC_LQR_HIP_LIB=variants/lib_w1_prof.so python tools/w1_prof.py
Round 3, pendulum / cart-pole: F'VF + q 496 / 976 cycles, broadcasts + scalar QP + gains 892 / 1316, gain stores 248 / 236,
V and v 140 / 204, a whole timestep 2156 / 3172 -- 8-10 cycles per instruction: one wavefront per SIMD on one dependent chain."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
from mpc import _native, mpc
from mpc.mpc import QuadCost
from mpc._native import StepOptions
from tools.bench_ilqr_env import problem
be = _native.HipBackend()
for kind, B, T in (("pendulum", 1024, 20), ("cartpole", 1024, 25)):
    dx, plain, x0, Q, pp = problem(kind, B, T)
    ctrl = mpc.MPC(dx.n_state, 1, T, u_lower=dx.lower, u_upper=dx.upper, lqr_iter=5, verbose=-1, exit_unconverged=False,
                   detach_unconverged=False, linesearch_decay=dx.linesearch_decay, max_linesearch_iter=dx.max_linesearch_iter,
                   grad_method=mpc.GradMethods.AUTO_DIFF, eps=1e-12, backprop=False, not_improved_lim=10 ** 6)
    x, u, _ = ctrl(x0, QuadCost(Q, pp), dx)
    env = dx.native_env(); env.linearize = True
    o = StepOptions(u_lower=dx.lower, u_upper=dx.upper, linesearch_decay=dx.linesearch_decay, max_linesearch_iter=dx.max_linesearch_iter, true_dynamics=env)
    plan = be.plan_step(x0, Q, pp, None, None, x.detach().contiguous(), u.detach().contiguous(), o, impl=6)
    for _ in range(3):
        r = plan(); torch.cuda.synchronize()
        print(kind, "cycles: FVF+q | bcast+QP+K | stores | M | V,v | rest of loop to next step:", r["qp_iters"][8:13].tolist(), "qp_iters sample", r["qp_iters"][:4].tolist())
