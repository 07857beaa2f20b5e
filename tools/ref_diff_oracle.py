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
def alone(cs, b):
    sl = slice(b, b + 1)
    cut = lambda v: (v[:, sl] if v.ndim >= 3 else v[sl]) if isinstance(v, np.ndarray) else v
    d = {k: cut(v) for k, v in cs.items() if k not in ("kw",)}
    d["kw"] = {k: cut(v) for k, v in cs["kw"].items()}
    d["B"] = 1; d["grads"] = False
    return d
singles = [alone(cs, b) for cs in cases for b in range(cs["B"])]
tmp = tempfile.mkdtemp()
pickle.dump(cases + singles, open(os.path.join(tmp, "cases.pkl"), "wb"))
env = dict(os.environ); env.pop("PYTHONPATH", None)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ref_diff_oracle_child.py"), os.path.join(tmp, "cases.pkl"), os.path.join(tmp, "ref.pkl")], env=env, cwd=tmp)
ref_all = pickle.load(open(os.path.join(tmp, "ref.pkl"), "rb"))
ref, ref1 = ref_all[:len(cases)], ref_all[len(cases):]
bad = ref_broken = coupled = 0
k1 = 0
for i, (cs, r) in enumerate(zip(cases, ref)):
    mine = ref1[k1:k1 + cs["B"]]; k1 += cs["B"]
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
    # THE contract the kernels are held to: every problem as the reference solves it ALONE (n_batch = 1) against lockstep=False
    if not any("error" in m1 for m1 in mine):
        op = O.lqr_step(cs["x_init"], cs["C"], cs["c"], cs["F"], cs["f"], cs["cur_x"], cs["cur_u"], kw.get("u_lower"), kw.get("u_upper"),
                        kw.get("u_zero_I"), kw.get("delta_u"), kw["linesearch_decay"], kw["max_linesearch_iter"], lockstep=False)
        alone_w = dict(x=max(rel(op["new_x"][:, b], m1["new_x"][:, 0]) for b, m1 in enumerate(mine)),
                       u=max(rel(op["new_u"][:, b], m1["new_u"][:, 0]) for b, m1 in enumerate(mine)),
                       costs=max(rel(op["costs"][b:b + 1], m1["costs"]) for b, m1 in enumerate(mine)),
                       n_qp=abs(float(np.sum(op["n_qp_iter"])) - sum(m1["n_qp"] for m1 in mine)))
        if max(v for k, v in alone_w.items() if k != "n_qp") > lim:          # (the iteration total is a batch quantity: compared in the batched run below)
            cyc = np.linalg.eigvalsh(cs["C"][:, :, :cs["ns"], :cs["ns"]]).min() < 0 and max(m1["n_qp"] for m1 in mine) >= 10 * cs["T"]
            if cyc:
                # a non-convex problem whose box QPs run into the iteration cap: twenty trips of a solver that cycles amplify the
                # last bit of every solve -- the reference's own answer moves with the LAPACK path torch takes (a batch of one
                # against a batch of four); nothing to hold a restatement to.  Named, counted.
                coupled += 1
                continue
            bad += 1
            print("VIOLATION (per problem) case %d ns %d nc %d T %d B %d mode %s: %s" % (i, cs["ns"], cs["nc"], cs["T"], cs["B"], cs["mode"], alone_w))
            continue
    nonconvex = np.linalg.eigvalsh(cs["C"][:, :, :cs["ns"], :cs["ns"]]).min() < 0
    if (max(v for k, v in worst.items() if k not in ("n_qp",)) > lim or worst["n_qp"] > 0) and nonconvex and cs["B"] > 1 and not cs["grads"]:
        # the BATCHED call of a non-convex problem: the reference's batch-global pnqp loop keeps iterating every problem while any
        # has not converged, and a box QP that cycles ends wherever the batch's count stops it; the restatement of that coupling
        # parts ways with it there (the per-problem contract above held).  Named, counted.
        coupled += 1
        continue
    if max(v for k, v in worst.items() if k != "n_qp") > lim or worst["n_qp"] > 0:
        bad += 1
        print("VIOLATION case %d ns %d nc %d T %d B %d mode %s kw %s grads %s: %s" % (i, cs["ns"], cs["nc"], cs["T"], cs["B"], cs["mode"],
              {k: (v if not isinstance(v, np.ndarray) else "array") for k, v in kw.items()}, cs["grads"], worst))
print("cases %d violations %d (the reference itself raised: %d; non-convex problems whose cycling box QPs part ways: %d)" % (n_cases, bad, ref_broken, coupled))
sys.exit(1 if bad else 0)
