#!/usr/bin/env python3
"""No GPU, BUILD CONTAINER ONLY: the CPU oracle (oracle/) against the UNMODIFIED reference's LQRStep on RANDOM problems -- the
pin of tests/test_oracle_golden.py (72 fixtures) widened to thousands of cases: forward (new_x, new_u, costs, pnqp iteration total,
the batch's du norm) with lockstep semantics, and LQRStepFn.backward through the reference's own autograd against kkt_backward.
Shapes, horizons, batches, bounds (none / scalar / tensor), delta_u, u_zero_I, f on / off, line-search depth, nominals on and off
their rollout, float64.      python tools/ref_diff_oracle.py [cases [seed]]"""
import os, pickle, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MPC_REFERENCE_DIR", "/root/reference")
if not os.path.isdir(os.path.join(REF, "mpc")):
    print("no reference under %s: nothing to compare with" % REF); sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import lqr_oracle as O
from helpers import scrambled_du_norm

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cases = []
for i in range(n_cases):
    rng = np.random.default_rng(seed0 * 999983 + i)
    ns, nc = int(rng.integers(1, 7)), int(rng.integers(1, 5))
    n = ns + nc
    T, B = int(rng.integers(1, 9)), int(rng.integers(1, 5))
    A = rng.standard_normal((T, B, n, n)); C = np.einsum("tbji,tbjk->tbik", A, A) + 0.1 * np.eye(n)
    if rng.random() < 0.15:
        C[:, :, :ns, :ns] -= float(rng.choice([5.0, 20.0])) * np.eye(ns)          # a non-convex state cost: the line search really backtracks
    c = rng.standard_normal((T, B, n))
    F = np.concatenate((np.eye(ns) + 0.2 * rng.standard_normal((max(T - 1, 0), B, ns, ns)), rng.standard_normal((max(T - 1, 0), B, ns, nc))), 3)
    f = 0.1 * rng.standard_normal((max(T - 1, 0), B, ns)) if (rng.random() < 0.7 and T > 1) else None
    x_init = rng.standard_normal((B, ns))
    kw = dict(linesearch_decay=float(rng.choice([0.2, 0.5])), max_linesearch_iter=int(rng.choice([1, 2, 5, 10])))
    mode = str(rng.choice(["none", "scalar", "tensor", "mask"], p=[0.25, 0.3, 0.3, 0.15]))
    bnd = float(rng.choice([0.25, 0.5, 1.0]))
    cur_u = np.clip(0.5 * rng.standard_normal((T, B, nc)), -bnd, bnd)
    if mode == "scalar":
        kw.update(u_lower=-bnd, u_upper=bnd)
    elif mode == "tensor":
        kw.update(u_lower=-bnd - rng.random((T, B, nc)), u_upper=bnd + rng.random((T, B, nc)))
    elif mode == "mask":
        kw.update(u_zero_I=(rng.random((T, B, nc)) < 0.3))
        cur_u = np.where(kw["u_zero_I"], 0.0, cur_u)
    if mode in ("scalar", "tensor") and rng.random() < 0.3:
        kw["delta_u"] = float(rng.choice([0.1, 0.5]))
    cur_x, _ = O.traj_cost(x_init, cur_u, F, f)
    if rng.random() < 0.15 and T > 2:
        cur_x = cur_x.copy(); cur_x[2:] += 0.05 * rng.standard_normal(cur_x[2:].shape)     # LQRStep allows a nominal off its rollout
    cases.append(dict(ns=ns, nc=nc, T=T, B=B, C=C, c=c, F=F, f=f, x_init=x_init, cur_x=cur_x, cur_u=cur_u, kw=kw, mode=mode,
                      grads=bool(rng.random() < 0.6 and mode != "mask"), wx=rng.standard_normal((T, B, ns)), wu=rng.standard_normal((T, B, nc))))
tmp = tempfile.mkdtemp()
pickle.dump(cases, open(os.path.join(tmp, "cases.pkl"), "wb"))
env = dict(os.environ); env.pop("PYTHONPATH", None)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ref_diff_oracle_child.py"), os.path.join(tmp, "cases.pkl"), os.path.join(tmp, "ref.pkl")], env=env, cwd=tmp)
ref = pickle.load(open(os.path.join(tmp, "ref.pkl"), "rb"))
bad = ref_broken = 0
for i, (cs, r) in enumerate(zip(cases, ref)):
    if "error" in r:
        ref_broken += 1
        if "masked_fill_" not in r["error"]:
            print("reference raised on case %d: %s" % (i, r["error"]))
        continue
    kw = cs["kw"]
    o = O.lqr_step(cs["x_init"], cs["C"], cs["c"], cs["F"], cs["f"], cs["cur_x"], cs["cur_u"], kw.get("u_lower"), kw.get("u_upper"),
                   kw.get("u_zero_I"), kw.get("delta_u"), kw["linesearch_decay"], kw["max_linesearch_iter"], lockstep=True)
    rel = lambda a, b: float("%.3g" % (np.abs(a - b).max() / max(1.0, np.abs(b).max()))) if a.size else 0.0
    worst = dict(x=rel(o["new_x"], r["new_x"]), u=rel(o["new_u"], r["new_u"]), costs=rel(o["costs"], r["costs"]),
                 n_qp=abs(float(np.sum(o["n_qp_iter"])) - r["n_qp"]) if "n_qp_iter" in o else 0.0)
    if cs["grads"]:
        g = O.kkt_backward(cs["C"], cs["c"], cs["F"], cs["f"], r["new_x"], r["new_u"], cs["wx"], cs["wu"], kw.get("u_lower"), kw.get("u_upper"), lockstep=True)
        for k in ("dC", "dc", "dF", "dx_init") + (("df",) if cs["f"] is not None else ()):
            worst[k] = rel(g[k], r[k])
    lim = 1e-7
    if max(v for k, v in worst.items() if k != "n_qp") > lim or worst["n_qp"] > 0:
        bad += 1
        print("VIOLATION case %d ns %d nc %d T %d B %d mode %s kw %s grads %s: %s" % (i, cs["ns"], cs["nc"], cs["T"], cs["B"], cs["mode"],
              {k: (v if not isinstance(v, np.ndarray) else "array") for k, v in kw.items()}, cs["grads"], worst))
print("cases %d violations %d (the reference itself raised: %d)" % (n_cases, bad, ref_broken))
sys.exit(1 if bad else 0)
