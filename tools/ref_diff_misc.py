#!/usr/bin/env python3
"""No GPU, BUILD CONTAINER ONLY: the oracle's other entry points against the UNMODIFIED reference on random inputs -- pnqp
(mpc/pnqp.py:5-82: solution, free set, iteration count, batched), util.get_traj / get_cost (mpc/util.py:102-153), NNDynamics
forward and grad_input (mpc/dynamics.py:57-128: the checker of the network kernels), the shipped simulators (mpc/env_dx: next state, and the
Jacobian + affine term MPC.linearize_dynamics(AUTO_DIFF) forms, controls past their clamp included).  float64.
    python tools/ref_diff_misc.py [cases [seed]]"""
import os, pickle, subprocess, sys, tempfile, types
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MPC_REFERENCE_DIR", "/root/reference")
if not os.path.isdir(os.path.join(REF, "mpc")):
    print("no reference under %s: nothing to compare with" % REF); sys.exit(0)
sys.path.insert(0, ROOT)
from oracle import lqr_oracle as O
from oracle import env_oracle as E

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cases = []
for i in range(n_cases):
    rng = np.random.default_rng(seed0 * 777767 + i)
    kind = ("pnqp", "traj", "nn", "env")[i % 4]
    if kind == "pnqp":
        B, n = int(rng.integers(1, 6)), int(rng.integers(1, 11))
        A = rng.standard_normal((B, n, n)); H = np.einsum("bji,bjk->bik", A, A) + 0.05 * np.eye(n)
        q = 3 * rng.standard_normal((B, n))
        lo = -rng.random((B, n)) - 0.05; hi = rng.random((B, n)) + 0.05
        x0 = None if rng.random() < 0.5 else rng.standard_normal((B, n))
        cases.append(dict(kind=kind, H=H, q=q, lo=lo, hi=hi, x0=x0, n_iter=int(rng.choice([20, 3, 1]))))
    elif kind == "traj":
        T, B, ns, nc = int(rng.integers(1, 10)), int(rng.integers(1, 5)), int(rng.integers(1, 7)), int(rng.integers(1, 4))
        n = ns + nc
        A = rng.standard_normal((T, B, n, n)); C = np.einsum("tbji,tbjk->tbik", A, A)
        cases.append(dict(kind=kind, T=T, C=C, c=rng.standard_normal((T, B, n)), F=rng.standard_normal((max(T - 1, 0), B, ns, n)),
                          f=rng.standard_normal((max(T - 1, 0), B, ns)) if (rng.random() < 0.7 and T > 1) else None,
                          x_init=rng.standard_normal((B, ns)), u=rng.standard_normal((T, B, nc))))
    elif kind == "env":
        env = str(rng.choice(["pendulum", "cartpole"])); N = int(rng.integers(1, 30))
        if env == "pendulum":
            simple = bool(rng.integers(0, 2))
            params = np.array([10.0, 1.0, 1.0]) * (0.5 + rng.random(3)) if simple else np.concatenate((np.array([10.0, 1.0, 1.0]) * (0.5 + rng.random(3)), 0.2 * rng.random(2)))
            th = (rng.random(N) - 0.5) * 2 * np.pi
            x = np.stack((np.cos(th), np.sin(th), 4 * (rng.random(N) - 0.5)), 1); u = 3.0 * rng.standard_normal((N, 1))     # (some past the +-2 clamp)
            cases.append(dict(kind=kind, env=env, simple=simple, params=params, x=x, u=u))
        else:
            params = np.array([9.8, 1.0, 0.1, 0.5]) * (0.5 + rng.random(4))
            th = (rng.random(N) - 0.5) * 2 * np.pi
            x = np.stack((rng.standard_normal(N), rng.standard_normal(N), np.cos(th), np.sin(th), 2 * rng.standard_normal(N)), 1)
            cases.append(dict(kind=kind, env=env, simple=True, params=params, x=x, u=60.0 * rng.standard_normal((N, 1))))        # (some past +-100)
    else:
        ns, nc = int(rng.integers(1, 17)), int(rng.integers(1, 9))
        hidden = [int(rng.integers(1, 120)) for _ in range(int(rng.integers(0, 4)))]
        sizes = [ns + nc] + hidden + [ns]
        Ws = [rng.standard_normal((sizes[k + 1], sizes[k])) / np.sqrt(sizes[k]) for k in range(len(sizes) - 1)]
        bs = [0.1 * rng.standard_normal(sizes[k + 1]) for k in range(len(sizes) - 1)]
        N = int(rng.integers(1, 40))
        cases.append(dict(kind=kind, ns=ns, nc=nc, hidden=hidden, act=str(rng.choice(["sigmoid", "relu"])), passthrough=bool(rng.integers(0, 2)),
                          Ws=Ws, bs=bs, x=rng.standard_normal((N, ns)), u=rng.standard_normal((N, nc))))
tmp = tempfile.mkdtemp()
pickle.dump(cases, open(os.path.join(tmp, "cases.pkl"), "wb"))
env = dict(os.environ); env.pop("PYTHONPATH", None)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ref_diff_misc_child.py"), os.path.join(tmp, "cases.pkl"), os.path.join(tmp, "ref.pkl")], env=env, cwd=tmp)
ref = pickle.load(open(os.path.join(tmp, "ref.pkl"), "rb"))
bad = broken = 0
rel = lambda a, b: float("%.3g" % (np.abs(a - b).max() / max(1.0, np.abs(b).max()))) if np.asarray(a).size else 0.0
for i, (cs, r) in enumerate(zip(cases, ref)):
    if "error" in r:
        broken += 1
        if "masked_fill_" not in r["error"] and "uint8" not in r["error"] and "Byte" not in r["error"]:
            print("reference raised on case %d (%s): %s" % (i, cs["kind"], r["error"]))
        continue
    if cs["kind"] == "pnqp":
        o = O.pnqp(cs["H"], cs["q"], cs["lo"], cs["hi"], x_init=cs["x0"], n_iter=cs["n_iter"], lockstep=True)
        w = dict(x=rel(o["x"], r["x"]), If=float(np.abs(o["If"].astype(float) - r["If"].astype(float)).max()), n=abs(int(o["iters"].max()) - r["n"]))
        ok = w["x"] < 1e-9 and w["If"] == 0 and w["n"] == 0
    elif cs["kind"] == "traj":
        x, cost = O.traj_cost(cs["x_init"], cs["u"], cs["F"], cs["f"], cs["C"], cs["c"])
        w = dict(x=rel(x, r["x"]), cost=rel(cost, r["cost"]))
        ok = max(w.values()) < 1e-10
    elif cs["kind"] == "env":
        kd = E.CARTPOLE if cs["env"] == "cartpole" else (E.PENDULUM if cs["simple"] else E.PENDULUM_FULL)
        y = E.step(kd, cs["x"], cs["u"], cs["params"])
        F, f = E.linearize(kd, cs["x"], cs["u"], cs["params"])
        w = dict(y=rel(y, r["y"]), J=rel(F, r["J"]), f=rel(f, r["y"] - np.einsum("nij,nj->ni", r["J"], np.concatenate((cs["x"], cs["u"]), 1))))
        ok = w["y"] < 1e-12 and w["J"] < 1e-9 and w["f"] < 1e-9
    else:
        net = types.SimpleNamespace(Ws=cs["Ws"], bs=cs["bs"], activation=cs["act"], passthrough=cs["passthrough"])
        w = dict(y=rel(E.mlp_step(cs["x"], cs["u"], net), r["y"]), J=rel(E.mlp_jacobian(cs["x"], cs["u"], net), r["J"]))
        ok = max(w.values()) < 1e-10
    if not ok:
        bad += 1
        print("VIOLATION case %d %s: %s" % (i, cs["kind"], w))
print("cases %d violations %d (the reference itself raised: %d)" % (n_cases, bad, broken))
sys.exit(1 if bad else 0)
