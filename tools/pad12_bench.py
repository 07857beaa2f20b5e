#!/usr/bin/env python3
"""Round 6: the padded 12/4 kernel (impl 8, lqr_dpp16_body.h PADK) against the one-problem-per-wavefront kernel (impl 2) that served every
shape <= 12/4 other than 12/4 in rounds 1-5, and against the exact kernel (impl 3) at 12/4: sustained time per step, fraction of each
shape's own roofline, parity of the first 64 problems against the oracle.   python tools/pad12_bench.py"""
import sys, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'mpc.pytorch_amd'))
import bench
from mpc import _native
from mpc._native import StepOptions
from oracle import lqr_oracle as O
be=_native.HipBackend()
def run(ns,nc,T,B,bounded,impl,n=20,check=True):
    p=bench.make_problem(ns,nc,T,B,torch.float32,"cuda:0",seed=3,u_scale=0.3 if bounded else 0.0,clamp=1.0 if bounded else None)
    opts=StepOptions(u_lower=-1.0,u_upper=1.0,nominal_on_dynamics=True,c_symmetric=True) if bounded else StepOptions(nominal_on_dynamics=True,c_symmetric=True)
    plan=be.plan_step(p["x_init"],p["C"],p["c"],p["F"],p["f"],p["cur_x"],p["cur_u"],opts,impl=impl)
    for _ in range(150): r=plan()
    _,ms,r=bench.timed(plan,40,0)
    err=None
    if check:
        m=min(B,64)
        h={k:(v[:,:m] if k!="x_init" else v[:m]).cpu().numpy().astype(np.float64) for k,v in p.items()}
        o=O.lqr_step(h["x_init"],h["C"],h["c"],h["F"],h["f"],h["cur_x"],h["cur_u"],-1.0 if bounded else None,1.0 if bounded else None,lockstep=False,qp_cold=True)
        err=max(float(np.abs(r["new_u"][:,:m].cpu().numpy()-o["new_u"]).max()),float(np.abs(r["new_x"][:,:m].cpu().numpy()-o["new_x"]).max()))
    ab=bench.algorithmic_bytes_per_problem(ns,nc,T)*B
    print("%2d/%d T=%d B=%d %s impl %d: %.1f us  frac %.3f  max err %s" % (ns,nc,T,B,"bounded" if bounded else "unbounded",impl,ms*1e3,ab/(ms*1e-3)/8e12,err),flush=True)
for bounded in (False,True):
    for (ns,nc) in ((12,4),(8,4),(10,3),(12,2),(6,2),(4,4)):
        for impl in ((3,8,2) if (ns,nc)==(12,4) else (8,2)):
            run(ns,nc,50,4096,bounded,impl)
