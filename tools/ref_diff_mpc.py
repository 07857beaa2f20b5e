#!/usr/bin/env python3
"""No GPU, BUILD CONTAINER ONLY (needs the reference under /root/reference): `mpc.MPC` of this package -- the host logic around
the kernels: iteration, best-iterate bookkeeping, convergence exits, detach_unconverged, the autograd wiring of LQRStep -- on the
oracle-backed stand-in (tests/oracle_backend.py, batch-lockstep like the reference) against the UNMODIFIED reference's `mpc.MPC`
on random configurations: shape, horizon, batch, bounds, lqr_iter, eps, not_improved_lim, best_cost_eps, delta_u, line-search
settings, u_init, exit_unconverged / detach_unconverged, gradients of a random functional of u w.r.t. C, c, x_init.  float64.
    python tools/ref_diff_mpc.py [cases [seed]]"""
import os, pickle, subprocess, sys, tempfile, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MPC_REFERENCE_DIR", "/root/reference")
if not os.path.isdir(os.path.join(REF, "mpc")):
    print("no reference under %s: nothing to compare with" % REF); sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd"))
import torch
from mpc import _native, mpc
from mpc.mpc import QuadCost, LinDx
from oracle_backend import OracleBackend
warnings.filterwarnings("ignore")

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cases = []
for i in range(n_cases):
    rng = np.random.default_rng(seed0 * 1000003 + i)
    ns, nc = int(rng.integers(1, 6)), int(rng.integers(1, 4))
    n = ns + nc
    T, B = int(rng.integers(2, 9)), (1 if rng.random() < 0.6 else int(rng.integers(2, 5)))
    A = rng.standard_normal((T, B, n, n)); C = np.einsum("tbji,tbjk->tbik", A, A) + 0.1 * np.eye(n)
    c = rng.standard_normal((T, B, n))
    F = np.concatenate((np.eye(ns) + 0.2 * rng.standard_normal((T - 1, B, ns, ns)), rng.standard_normal((T - 1, B, ns, nc))), 3)
    f = 0.1 * rng.standard_normal((T - 1, B, ns)) if rng.random() < 0.7 else None
    kw = dict(lqr_iter=int(rng.choice([1, 2, 3, 5, 10])), exit_unconverged=False, backprop=True,
              eps=float(rng.choice([1e-7, 1e-4, 1e-2])), not_improved_lim=int(rng.choice([1, 2, 5])),
              best_cost_eps=float(rng.choice([1e-4, 1e-8, 1e-2])), linesearch_decay=float(rng.choice([0.2, 0.5])),
              max_linesearch_iter=int(rng.choice([1, 3, 10])), detach_unconverged=bool(rng.integers(0, 2)))
    mode = str(rng.choice(["none", "scalar", "tensor"]))
    if mode == "scalar":
        b = float(rng.choice([0.25, 0.5, 1.0])); kw.update(u_lower=-b, u_upper=b)
    elif mode == "tensor":
        kw.update(u_lower=-0.5 - rng.random((T, B, nc)), u_upper=0.5 + rng.random((T, B, nc)))
    if mode != "none" and rng.random() < 0.3:
        kw["delta_u"] = float(rng.choice([0.1, 0.5]))
    if rng.random() < 0.3:
        kw["u_init"] = 0.1 * rng.standard_normal((T, B, nc))
    if rng.random() < 0.15:
        kw["exit_unconverged"] = True
    if rng.random() < 0.2:
        kw["u_zero_I"] = (rng.random((T, B, nc)) < 0.3)
    # (slew_rate_penalty: the reference's own path for it runs on module dynamics only -- with LinDx it raises 'NoneType is not callable';
    #  the goldens mpc_slew_* made from tests/test_mpc.py:652-744 cover it)
    ref_norm = False
    if B > 1:
        # The reference's full_du_norm mixes the problems of a batch (mpc/lqr_step.py:243-245: transpose(1,2).view(n_batch, -1));
        # this package's is per problem (= the reference at n_batch = 1).  Where that number decides -- the eps exit and
        # which problems detach_unconverged cuts off -- a batch is comparable only with `reference_du_norm=True` (round 6), which every
        # second batch runs with; the others run with neither the exit nor the mask in play.
        if rng.random() < 0.5:
            ref_norm = True
        else:
            kw.update(eps=1e-13, detach_unconverged=False)
    grads = bool(rng.random() < 0.6)
    if rng.random() < 0.25:
        # iLQR on a shipped simulator (mpc/env_dx): time-invariant goal cost, the module linearised every iteration and rolled out in
        # the line search; this package's modules take the closed-form-Jacobian / in-kernel-rollout route (here: their oracle stand-ins).
        # (AUTO_DIFF only: the reference's FINITE_DIFF route raises 'batch1 must be a 3D tensor' on these modules under torch 2.x)
        env = str(rng.choice(["pendulum", "cartpole"]))
        ns, nc = (3, 1) if env == "pendulum" else (5, 1)
        T, B = int(rng.integers(3, 9)), int(rng.integers(1, 4))
        simple = bool(rng.integers(0, 2))
        if env == "pendulum":
            params = np.array([10.0, 1.0, 1.0]) * (0.7 + 0.6 * rng.random(3)) if simple else np.concatenate((np.array([10.0, 1.0, 1.0]) * (0.7 + 0.6 * rng.random(3)), 0.2 * rng.random(2)))
            th = (rng.random(B) - 0.5) * np.pi
            x_init = np.stack((np.cos(th), np.sin(th), 2 * (rng.random(B) - 0.5)), 1)
            goal, wts, bnd = np.array([1.0, 0.0, 0.0]), np.array([1.0, 1.0, 0.1]), 2.0
        else:
            params = np.array([9.8, 1.0, 0.1, 0.5]) * (0.7 + 0.6 * rng.random(4))
            th = (rng.random(B) - 0.5) * 1.0
            x_init = np.stack((0.5 * rng.standard_normal(B), 0.5 * rng.standard_normal(B), np.cos(th), np.sin(th), 0.5 * rng.standard_normal(B)), 1)
            goal, wts, bnd = np.array([0.0, 0.0, 1.0, 0.0, 0.0]), np.array([0.1, 0.1, 1.0, 1.0, 0.1]), 100.0
        q = np.concatenate((wts, 0.001 * np.ones(nc)))
        Cq = np.broadcast_to(np.diag(q), (T, B, ns + nc, ns + nc)).copy()
        cq = np.broadcast_to(-np.concatenate((np.sqrt(wts) * goal, np.zeros(nc))) * np.sqrt(q), (T, B, ns + nc)).copy()
        kw = dict(lqr_iter=int(rng.choice([1, 2, 4, 6])), exit_unconverged=False, detach_unconverged=False, backprop=False, eps=1e-13,
                  u_lower=-bnd, u_upper=bnd, linesearch_decay=float(rng.choice([0.2, 0.5])), max_linesearch_iter=int(rng.choice([3, 5, 10])),
                  grad_method="AUTO_DIFF", not_improved_lim=5, best_cost_eps=1e-4)
        cases.append(dict(ns=ns, nc=nc, T=T, B=B, C=Cq, c=cq, F=None, f=None, x_init=x_init, kw=kw, grads=False, w=None,
                          env=env, params=params, simple=simple))
        continue
    if rng.random() < 0.12:
        # iLQR on NNDynamics (mpc/dynamics.py): the network linearised by its own grad_input (ANALYTIC) every iteration
        hidden = [int(rng.integers(2, 24)) for _ in range(int(rng.integers(1, 3)))]
        sizes = [n] + hidden + [ns]
        net = dict(hidden=hidden, act=str(rng.choice(["sigmoid", "relu"])), passthrough=bool(rng.integers(0, 2)),
                   Ws=[rng.standard_normal((sizes[k + 1], sizes[k])) / np.sqrt(sizes[k]) for k in range(len(sizes) - 1)],
                   bs=[0.1 * rng.standard_normal(sizes[k + 1]) for k in range(len(sizes) - 1)])
        kwn = dict(lqr_iter=int(rng.choice([1, 2, 4])), exit_unconverged=False, detach_unconverged=False, backprop=False, eps=1e-13,
                   u_lower=-1.0, u_upper=1.0, linesearch_decay=float(rng.choice([0.2, 0.5])), max_linesearch_iter=int(rng.choice([3, 10])),
                   grad_method="ANALYTIC", not_improved_lim=5, best_cost_eps=1e-4)
        cases.append(dict(ns=ns, nc=nc, T=T, B=B, C=C, c=c, F=None, f=None, x_init=rng.standard_normal((B, ns)), kw=kwn, grads=False, w=None, net=net))
        continue
    cases.append(dict(ns=ns, nc=nc, T=T, B=B, C=C, c=c, F=F, f=f, x_init=rng.standard_normal((B, ns)), kw=kw, grads=grads, ref_norm=ref_norm,
                      w=rng.standard_normal((T, B, nc))))
tmp = tempfile.mkdtemp()
pickle.dump(cases, open(os.path.join(tmp, "cases.pkl"), "wb"))
env = dict(os.environ); env.pop("PYTHONPATH", None)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ref_diff_child.py"), os.path.join(tmp, "cases.pkl"), os.path.join(tmp, "ref.pkl")], env=env, cwd=tmp)
ref = pickle.load(open(os.path.join(tmp, "ref.pkl"), "rb"))

prev = _native.set_backend_for_testing(OracleBackend(lockstep=True))
bad = errors_both = 0
try:
    for i, (cs, r) in enumerate(zip(cases, ref)):
        t = lambda a: None if a is None else torch.from_numpy(a).clone()
        C, c, F, f, x0 = t(cs["C"]), t(cs["c"]), t(cs["F"]), t(cs["f"]), t(cs["x_init"])
        if cs["grads"]:
            C.requires_grad_(True); c.requires_grad_(True); x0.requires_grad_(True)
        kw = {k: (t(v) if isinstance(v, np.ndarray) else v) for k, v in cs["kw"].items()}
        try:
            dyn = None
            if cs.get("env"):
                from mpc.env_dx import cartpole, pendulum
                dyn = pendulum.PendulumDx(params=t(cs["params"]), simple=cs["simple"]) if cs["env"] == "pendulum" else cartpole.CartpoleDx(params=t(cs["params"]))
                kw["grad_method"] = getattr(mpc.GradMethods, kw["grad_method"])
            if cs.get("net"):
                from mpc.dynamics import NNDynamics
                nt = cs["net"]
                dyn = NNDynamics(cs["ns"], cs["nc"], hidden_sizes=list(nt["hidden"]), activation=nt["act"], passthrough=nt["passthrough"]).double()
                with torch.no_grad():
                    for fc, W, b in zip(dyn.fcs, nt["Ws"], nt["bs"]):
                        fc.weight.copy_(t(W)); fc.bias.copy_(t(b))
                kw["grad_method"] = getattr(mpc.GradMethods, kw["grad_method"])
            ctrl = mpc.MPC(cs["ns"], cs["nc"], cs["T"], verbose=-1, reference_du_norm=bool(cs.get("ref_norm")), **kw)
            x, u, costs = ctrl(x0, QuadCost(C, c), dyn if dyn is not None else LinDx(F, f))
            m = dict(x=x.detach().numpy(), u=u.detach().numpy(), costs=costs.detach().numpy())
            if cs["grads"]:
                (u * torch.from_numpy(cs["w"])).sum().backward()
                m.update(gC=C.grad.numpy(), gc=c.grad.numpy(), gx0=x0.grad.numpy())
        except Exception as e:
            m = dict(error=type(e).__name__ + ": " + str(e)[:200])
        if "error" in r and "masked_fill_ only supports boolean masks" in r["error"]:
            ref_broken = globals().get("ref_broken", 0) + 1; globals()["ref_broken"] = ref_broken      # (the reference's uint8 masks under torch 2.x: nothing to compare)
            continue
        if "error" in r or "error" in m:
            if ("error" in r) != ("error" in m) and cs["kw"].get("exit_unconverged") and ("AssertionError" in r.get("error", "") or "UnconvergedError" in m.get("error", "")):
                # exit_unconverged: max ||du|| against eps at the last iterate.  There the box QP's answer is good to its own stopping
                # rule (|step| < 1e-4) and decided by exact `x == bound` tests: an input that differs in the 16th digit (the mirror
                # hands the step's own rollout on as the next nominal, the reference recomputes it) lands 1e-5 away.  Counted.
                edge = globals().get("edge", 0) + 1; globals()["edge"] = edge
                continue
            if ("error" in r) != ("error" in m):
                bad += 1
                print("VIOLATION case %d: reference %s | mirror %s | kw %s" % (i, r.get("error", "ok"), m.get("error", "ok"), {k: (v if not isinstance(v, np.ndarray) else "array") for k, v in cs["kw"].items()}))
            else:
                errors_both += 1
            continue
        worst = {}
        for k in r:
            worst[k] = float("%.3g" % (np.abs(m[k] - r[k]).max() / max(1.0, np.abs(r[k]).max())))
        if cs.get("env") or cs.get("net"):
            if (cs["kw"].get("u_lower") is not None and max(worst.values()) > 1e-6 and worst["costs"] < 1e-7 and max(worst["x"], worst["u"]) < 2e-4):
                # (round 6: the same edge as below, on a box-constrained iLQR solve -- costs equal to 1e-9, controls 1e-5 apart: a last
                # iterate inside pnqp's own stopping tolerance; counted against the same cap)
                edge = globals().get("edge", 0) + 1; globals()["edge"] = edge
                continue
            if max(worst.values()) > (2e-4 if cs["kw"]["grad_method"] == "FINITE_DIFF" else 1e-6):
                bad += 1
                print("VIOLATION (iLQR) case %d %s simple %s T %d B %d kw %s: %s" % (i, cs.get("env", "network"), cs.get("simple"), cs["T"], cs["B"], {k: v for k, v in cs["kw"].items()}, worst))
            continue
        if max(worst.values()) > 1e-6 and worst["costs"] < 1e-7 and max(worst["x"], worst["u"]) < 2e-4 and max(worst.values()) <= 10 * max(worst["x"], worst["u"]):
            edge = globals().get("edge", 0) + 1; globals()["edge"] = edge       # (the same: a last iterate inside the QP's own tolerance)
            continue
        if max(worst.values()) > 1e-6:
            bad += 1
            print("VIOLATION case %d ns %d nc %d T %d B %d grads %s kw %s: %s" % (i, cs["ns"], cs["nc"], cs["T"], cs["B"], cs["grads"],
                  {k: (v if not isinstance(v, np.ndarray) else "array") for k, v in cs["kw"].items()}, worst))
finally:
    _native.set_backend_for_testing(prev)
print("cases %d violations %d (both raised: %d; the reference itself crashed under this torch: %d; last iterates inside the QP's tolerance: %d)" % (n_cases, bad, errors_both, globals().get("ref_broken", 0), globals().get("edge", 0)))
if globals().get("edge", 0) > max(2, n_cases // 200):
    bad += 1
sys.exit(1 if bad else 0)
