#!/usr/bin/env python3
"""Child of tools/ref_diff_mpc.py: runs the UNMODIFIED reference (locuslab/mpc.pytorch under $MPC_REFERENCE_DIR or /root/reference)
on pickled cases and pickles what it returned.  Imports nothing of this repository (both packages are called `mpc`)."""
import os, pickle, sys, warnings
ref = os.environ.get("MPC_REFERENCE_DIR", "/root/reference")
sys.path.insert(0, ref)
import torch
from mpc import mpc
from mpc.mpc import QuadCost, LinDx
warnings.filterwarnings("ignore")
cases = pickle.load(open(sys.argv[1], "rb"))
out = []
for cs in cases:
    t = lambda a: None if a is None else torch.from_numpy(a).clone()
    C, c, F, f, x0 = t(cs["C"]), t(cs["c"]), t(cs["F"]), t(cs["f"]), t(cs["x_init"])
    if cs["grads"]:
        C.requires_grad_(True); c.requires_grad_(True); x0.requires_grad_(True)
    kw = dict(cs["kw"])
    for k in ("u_lower", "u_upper", "u_init", "u_zero_I"):
        if isinstance(kw.get(k), __import__("numpy").ndarray):
            kw[k] = t(kw[k])
    try:
        import io, contextlib, importlib
        dyn = None
        if cs.get("env"):                                    # iLQR on a shipped simulator: the module is the dynamics
            mod = importlib.import_module("mpc.env_dx." + cs["env"])
            dyn = mod.PendulumDx(params=t(cs["params"]), simple=cs["simple"]) if cs["env"] == "pendulum" else mod.CartpoleDx(params=t(cs["params"]))
            kw["grad_method"] = getattr(mpc.GradMethods, kw["grad_method"])
        if cs.get("net"):                                    # iLQR on NNDynamics: ANALYTIC linearisation through grad_input
            from mpc.dynamics import NNDynamics
            nt = cs["net"]
            dyn = NNDynamics(cs["ns"], cs["nc"], hidden_sizes=list(nt["hidden"]), activation=nt["act"], passthrough=nt["passthrough"]).double()
            with torch.no_grad():
                for fc, W, b in zip(dyn.fcs, nt["Ws"], nt["bs"]):
                    fc.weight.copy_(t(W)); fc.bias.copy_(t(b))
            kw["grad_method"] = getattr(mpc.GradMethods, kw["grad_method"])
        with contextlib.redirect_stdout(io.StringIO()):
            ctrl = mpc.MPC(cs["ns"], cs["nc"], cs["T"], verbose=-1, **kw)
            x, u, costs = ctrl(x0, QuadCost(C, c), dyn if dyn is not None else LinDx(F, f))
        r = dict(x=x.detach().numpy(), u=u.detach().numpy(), costs=costs.detach().numpy())
        if cs["grads"]:
            w = torch.from_numpy(cs["w"])
            (u * w).sum().backward()
            r.update(gC=C.grad.numpy(), gc=c.grad.numpy(), gx0=x0.grad.numpy())
    except Exception as e:                       # (the reference asserts / raises on some configurations: the mirror must raise too)
        r = dict(error=type(e).__name__ + ": " + str(e)[:200])
    out.append(r)
pickle.dump(out, open(sys.argv[2], "wb"))
