#!/bin/bash
# KKT backward at the headline shape: the product against a library variant (e.g. the KKT kernel on a 2-slot ring)
for lib in default "$@"; do
  if [ "$lib" != default ]; then export MPC_LQR_HIP_LIB=$PWD/$lib; else unset MPC_LQR_HIP_LIB; fi
  python - "$lib" <<'PY'
import sys, torch
sys.path.insert(0, "mpc.pytorch_amd"); sys.path.insert(0, ".")
import bench
from mpc import _native
from mpc._native import StepOptions
be = _native.backend()
for B in (4096, 8192):
    p = bench.make_problem(12, 4, 50, B, torch.float32, "cuda:0", seed=5)
    r = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], StepOptions(nominal_on_dynamics=True))
    gx, gu = torch.randn_like(r["new_x"]), torch.randn_like(r["new_u"])
    nx, nu = r["new_x"].clone(), r["new_u"].clone()
    w, ms, g = bench.timed(lambda: be.kkt_backward(p["C"], p["c"], p["F"], p["f"], nx, nu, gx, gu, StepOptions()), 30, 10)
    print(sys.argv[1], "kkt_backward B", B, "ms", round(ms, 4), "finite", bool(torch.isfinite(g["dC"]).all()))
PY
done
