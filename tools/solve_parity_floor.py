#!/usr/bin/env python3
"""How far apart do a float32 and a float64 run of the SAME whole-solve code end?  (bench.py's solve_parity gates the
simulator iLQR rows at rtol / atol 2e-2 on x, u and cited a measurement that was not in the tree.)
No GPU: mpc.MPC (this package's host logic) on tests/oracle_backend.py -- the checker of bench.py's `cfg2/cfg3_ilqr_*` rows -- solves
the first problems of those rows once in float32 and once in float64; the difference is what ANY float32 solve of the row owes the
float64 check before it has made a single error of its own.
    python tools/solve_parity_floor.py [n_problems]          log: profiles/r06_solve_parity_floor.log"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mpc import _native, mpc                      # noqa: E402
from mpc.mpc import QuadCost                      # noqa: E402
from oracle_backend import OracleBackend          # noqa: E402
import tools.bench_ilqr_env as _bie                          # noqa: E402
_bie.DEV = "cpu"
env_problem = _bie.problem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
_native.set_backend_for_testing(OracleBackend())
for kind, T in (("pendulum", 20), ("cartpole", 25)):
    dxm, _plain, x0, Q, pp = env_problem(kind, n, T)
    outs = {}
    for dt in (torch.float32, torch.float64):
        import copy
        d = copy.deepcopy(dxm)
        d.params = d.params.to(dt)
        ctrl = mpc.MPC(d.n_state, 1, T, u_lower=d.lower, u_upper=d.upper, lqr_iter=10, verbose=-1, exit_unconverged=False,
                       detach_unconverged=False, linesearch_decay=d.linesearch_decay, max_linesearch_iter=d.max_linesearch_iter,
                       grad_method=mpc.GradMethods.AUTO_DIFF, eps=1e-12, backprop=False, not_improved_lim=10 ** 6)
        with torch.no_grad():
            outs[dt] = [t.double().numpy() for t in ctrl(x0.to(dt), QuadCost(Q.to(dt), pp.to(dt)), d)]
    a, b = outs[torch.float32], outs[torch.float64]
    print("%-9s n=%d T=%d 10 iterations: max |dx| %.3g  max |du| %.3g  max relative cost difference %.3g  (|u| <= %.3g)" % (
        kind, n, T, np.abs(a[0] - b[0]).max(), np.abs(a[1] - b[1]).max(), (np.abs(a[2] - b[2]) / np.abs(b[2])).max(), float(dxm.upper)))
