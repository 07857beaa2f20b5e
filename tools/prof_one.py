#!/usr/bin/env python3
"""ONE kind of call, launched `reps` times behind `warm` launches of the same kind -- the command tools/prof_any.sh profiles,
so that a per-kernel average in profiles/r04_prof_*.json is the average of exactly that call kind (the
round-3 summaries mixed sweep-only, vouched and bare calls of one kernel in one average).

    python tools/prof_one.py KIND [reps [warm]]
    KIND: headline | headline_alt (two problem sets, alternating: bench.py's timed region) | headline_bare | bounded | kkt | kkt_bounded | cfg5 | cfg5_bare | cfg5_bounded | cfg5_kkt | cfg5_kkt_bounded |
          cfg5_B8192 | cfg5_bounded_B8192 | bounded_warm | cfg5_bounded_warm   (_warm: the box QPs started from the k an earlier
          step at the same nominal left in the workspace, mpc_lqr_options.qp_start) | pad12_<ns>_<nc>[_bounded] (round 6: the padded 12/4 kernel)
The problems are bench.py's (same seeds, same options as the rows of its `extra` object)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from mpc import _native
from mpc._native import StepOptions

kind = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 20
be = _native.HipBackend()
dev = "cuda:0"
cfg5 = kind.startswith("cfg5")
bounded = "bounded" in kind
bare = kind.endswith("_bare")
kind_b = kind[:-5] if kind.endswith("_warm") else kind
ns, nc, T = (32, 8, 64) if cfg5 else (12, 4, 50)
if kind.startswith("pad12_"):          # (round 6) pad12_<ns>_<nc>[_bounded]: a shape up to 12/4 on the padded 12/4 kernel (impl 0 = impl 8)
    ns, nc = int(kind.split("_")[1]), int(kind.split("_")[2])
B = 8192 if kind_b.endswith("B8192") else (1024 if cfg5 else 4096)
B = int(os.environ.get("PROF_ONE_B", B))          # (another batch for the same kind of call)
if cfg5:
    p = bench.make_problem(ns, nc, T, B, torch.float32, dev, seed=9, on_device=True)
else:
    p = bench.make_problem(ns, nc, T, B, torch.float32, dev, seed=5, u_scale=0.3 if bounded else 0.0, clamp=1.0 if bounded else None)
kw = dict(u_lower=-1.0, u_upper=1.0) if bounded else {}
if os.environ.get("PROF_ONE_MAXLS"):            # (what the line search beyond the first trial costs: 1 = full step only)
    kw["max_linesearch_iter"] = int(os.environ["PROF_ONE_MAXLS"])
opts = StepOptions(**kw) if bare else StepOptions(nominal_on_dynamics=True, c_symmetric=True, **kw)
a = (p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"])
if "kkt" in kind:
    r = be.lqr_step(*a, opts)
    nx, nu = r["new_x"].clone(), r["new_u"].clone()
    g = torch.Generator(device=dev).manual_seed(1)
    gx, gu = torch.randn(nx.shape, generator=g, device=dev), torch.randn(nu.shape, generator=g, device=dev)
    ko = StepOptions(c_symmetric=True, **kw)
    fn = be.plan_kkt_backward(p["C"], p["c"], p["F"], p["f"], nx, nu, gx, gu, ko)
elif kind == "headline_alt":
    # bench.py's timed region: the launches alternate between TWO problem sets (no launch finds its own inputs in the Infinity Cache)
    p2 = bench.make_problem(ns, nc, T, B, torch.float32, dev, seed=5 + 7919)
    plans = [be.plan_step(*a, opts), be.plan_step(p2["x_init"], p2["C"], p2["c"], p2["F"], p2["f"], p2["cur_x"], p2["cur_u"], opts)]
    turn = [0]

    def fn():
        turn[0] ^= 1
        return plans[turn[0]]()
else:
    fn = be.plan_step(*a, opts)
    if kind.endswith("_warm"):
        fn()                                        # the cold step leaves its k in the workspace
        import copy
        ow = copy.copy(opts)
        ow.qp_start = be.qp_record(fn)
        assert ow.qp_start is not None
        fn = be.plan_variant(fn, opts=ow)
if os.environ.get("PROF_ONE_TRACE"):
    # the launch-by-launch picture: ms per launch over consecutive groups of 20 launches from a GPU that has just idled through
    # the problem's set-up (boost clocks -> the power controller's dip -> the sustained state)
    torch.cuda.synchronize()
    groups = []
    for g_ in range(int(os.environ["PROF_ONE_TRACE"])):
        _, ms, _ = bench.timed(fn, 20, 0)
        groups.append(round(ms * 1e3, 1))
    print("prof_one %s: us per launch, consecutive groups of 20 launches: %s" % (kind, groups))
    sys.exit(0)
for _ in range(warm):
    fn()
torch.cuda.synchronize()
_, ms, _ = bench.timed(fn, reps, 0)
print("prof_one %s: B=%d ns=%d nc=%d T=%d reps=%d warm=%d  %.4f ms per call by HIP events" % (kind, B, ns, nc, T, reps, warm, ms))
