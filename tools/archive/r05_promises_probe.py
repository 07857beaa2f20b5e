#!/usr/bin/env python3
"""What each of the two promises is worth at config 5 and at the headline shape: nominal_on_dynamics x c_symmetric, HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import torch, bench
from mpc import _native
from mpc._native import StepOptions
be = _native.HipBackend()
dev = "cuda:0"
for (ns, nc, T, B, seed) in ((32, 8, 64, 1024, 9), (12, 4, 50, 4096, 5)):
    p = bench.make_problem(ns, nc, T, B, torch.float32, dev, seed=seed)
    a = (p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"])
    for nod in (True, False):
        for sym in (True, False):
            fn = be.plan_step(*a, StepOptions(nominal_on_dynamics=nod, c_symmetric=sym))
            for _ in range(200): fn()
            torch.cuda.synchronize()
            _, ms, _ = bench.timed(fn, 60, 0)
            print("%d/%d B=%d nominal_on_dynamics=%s c_symmetric=%s  %.1f us" % (ns, nc, B, nod, sym, ms * 1e3))
