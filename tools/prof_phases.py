#!/usr/bin/env python3
"""Where the cycles of the headline kernel go, phase by phase (diagnostic build -DMPC_DPP16_PROF: variants/prof.so).
Per wave: shader clocks spent in  DMA wait | LDS reads of the stage | DMA issue + pointer moves | arithmetic  of the sweep
and of the rollout.  Every probe drains the LDS / scalar queues, so the phases are serialised: upper bounds."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import bench
from mpc import _native
from mpc._native import StepOptions
be = _native.HipBackend()
out = {}
for B in (1024, 4096):
    for bounded in (False, True):
        p = bench.make_problem(12, 4, 50, B, torch.float32, "cuda:0", seed=1000, u_scale=0.3 if bounded else 0.0, clamp=1.0 if bounded else None)
        opts = StepOptions(u_lower=-1.0, u_upper=1.0) if bounded else StepOptions()
        for _ in range(20):
            r = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], opts, impl=3, want_gains=True)
        torch.cuda.synchronize()
        prof = r["K"].reshape(-1)[: (B // 4) * 16].reshape(B // 4, 16).cpu().numpy().astype(np.float64)
        names = ["sw_wait", "sw_lds", "sw_issue", "sw_rest(K,V,records)", "ro_wait", "ro_lds", "ro_issue", "ro_math", "setup+sweep_tail", "turn+rollout_tail",
                 "sw_products", "sw_control_block(QP)", "copy_trial2"]
        tot = prof[:, :13].sum(1)
        row = {n: round(float(prof[:, i].mean())) for i, n in enumerate(names)}
        row["total_mean"] = round(float(tot.mean())); row["total_max"] = round(float(tot.max())); row["total_min"] = round(float(tot.min()))
        out["B%d_%s" % (B, "bounded" if bounded else "unbounded")] = row
        print(B, bounded, json.dumps(row))
        # which phases make the slow waves slow (kernel time = slowest wave at one wave per SIMD)
        if B == 4096:
            order = np.argsort(tot)
            for label, idx in (("fastest 5%", order[: len(order) // 20]), ("median 10%", order[len(order) * 9 // 20: len(order) * 11 // 20]), ("slowest 5%", order[-(len(order) // 20):])):
                print("   ", label, {n: round(float(prof[idx, i].mean())) for i, n in enumerate(names)}, "total", round(float(tot[idx].mean())))
            xcd = np.arange(B // 4) % 8
            print("    mean total by XCD (block % 8):", [round(float(tot[xcd == x].mean())) for x in range(8)])

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "prof_phases.json"), "w"), indent=1)
