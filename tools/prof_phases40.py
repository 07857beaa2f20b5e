#!/usr/bin/env python3
"""Where the cycles of the 32/8 step kernel go, phase by phase (diagnostic build -DMPC_MFMA40_PROF: variants/prof40.so,
MPC_LQR_HIP_LIB).  Per wave (= per problem): shader clocks of the sweep's phases and of the rollout as a whole; every probe
drains the LDS queue, so the phases are serialised -- upper bounds.  Problems and options are bench.py's cfg5 rows."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import bench
from mpc import _native
from mpc._native import StepOptions
be = _native.HipBackend()
names = ["sw_dma_wait", "sw_C_tau_cback", "sw_F_products", "sw_control_block(QP)", "sw_Ksolve_M_stores", "sw_value_update", "sweep_tail",
         "rollout_rest", "sw_dma_issue", "ro_dma_wait", "ro_reads+issue", "ro_Kdx_Mdx", "ro_u_e_store", "ro_F_reads_x+", "ro_tail"]
out = {}
B, ns, nc, T = 1024, 32, 8, 64
p = bench.make_problem(ns, nc, T, B, torch.float32, "cuda:0", seed=9, on_device=True)
a = (p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"])
for bounded in (False, True):
    kw = dict(u_lower=-1.0, u_upper=1.0) if bounded else {}
    opts = StepOptions(nominal_on_dynamics=True, c_symmetric=True, **kw)
    for _ in range(30):
        r = be.lqr_step(*a, opts, impl=_native.IMPL_MFMA40, want_gains=True)
    torch.cuda.synchronize()
    prof = r["K"].reshape(-1)[: B * 16].reshape(B, 16).cpu().numpy().astype(np.float64)
    tot = prof[:, :15].sum(1)
    row = {n: round(float(prof[:, i].mean())) for i, n in enumerate(names)}
    row["total_mean"] = round(float(tot.mean())); row["total_max"] = round(float(tot.max())); row["total_min"] = round(float(tot.min()))
    out["bounded" if bounded else "unbounded"] = row
    print("bounded" if bounded else "unbounded", json.dumps(row))
    order = np.argsort(tot)
    for label, idx in (("fastest 5%", order[: len(order) // 20]), ("slowest 5%", order[-(len(order) // 20):])):
        print("   ", label, {n: round(float(prof[idx, i].mean())) for i, n in enumerate(names)}, "total", round(float(tot[idx].mean())))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "prof_phases40.json"), "w"), indent=1)
