#!/usr/bin/env python3
"""KKT backward at the headline shape: the one-launch route (mpc_lqr_kkt_fused, C vouched symmetric) against the
three-launch one (mpc_lqr_kkt_prepare + mpc_lqr_step + mpc_lqr_kkt_grads), interleaved on one box.
usage: python tools/ab_kkt.py [reps [ns nc T B]]      (32 8 64 1024: config 5, the fused route = nested step with the costates + outer products)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from mpc import _native
from mpc._native import StepOptions

be = _native.HipBackend()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
NS, NC, T, B = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (12, 4, 50, 4096)
res = {}
for bounded in (False, True):
    p = bench.make_problem(NS, NC, T, B, torch.float32, "cuda:0", seed=5, u_scale=0.3 if bounded else 0.0, clamp=1.0 if bounded else None)
    kw = dict(u_lower=-1.0, u_upper=1.0) if bounded else {}
    r = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], StepOptions(**kw))
    gx, gu = torch.randn_like(r["new_x"]), torch.randn_like(r["new_u"])
    nx, nu = r["new_x"].clone(), r["new_u"].clone()
    for name, o in (("fused", StepOptions(c_symmetric=True, **kw)), ("three_launch", StepOptions(**kw))):
        res.setdefault(("bounded_" if bounded else "unbounded_") + name, [])
    for _ in range(reps):
        for name, o in (("fused", StepOptions(c_symmetric=True, **kw)), ("three_launch", StepOptions(**kw))):
            _, ms, g = bench.timed(lambda: be.kkt_backward(p["C"], p["c"], p["F"], p["f"], nx, nu, gx, gu, o), 30, 8)
            res[("bounded_" if bounded else "unbounded_") + name].append(round(ms * 1e3, 1))
abytes = bench.kkt_algorithmic_bytes_per_problem(NS, NC, T) * B
print(json.dumps({"shape": [NS, NC, T, B], "us_per_backward": res, "algorithmic_MB": abytes / 1e6,
                  "frac_of_8TBs": {k: round(abytes / (min(v) * 1e-6) / 8e12, 3) for k, v in res.items()}}))
