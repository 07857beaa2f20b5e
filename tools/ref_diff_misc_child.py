#!/usr/bin/env python3
"""Child of tools/ref_diff_misc.py: the UNMODIFIED reference's pnqp, util.get_traj / get_cost and NNDynamics on pickled cases."""
import os, pickle, sys, warnings
sys.path.insert(0, os.environ.get("MPC_REFERENCE_DIR", "/root/reference"))
import numpy as np, torch
from mpc import mpc as _m, util
from mpc.pnqp import pnqp
from mpc.mpc import QuadCost, LinDx
from mpc.dynamics import NNDynamics
warnings.filterwarnings("ignore")
cases = pickle.load(open(sys.argv[1], "rb"))
out = []
t = lambda a: None if a is None else torch.from_numpy(a).clone()
for cs in cases:
    try:
        if cs["kind"] == "pnqp":
            x, H_lu, If, n = pnqp(t(cs["H"]), t(cs["q"]), t(cs["lo"]), t(cs["hi"]), x_init=t(cs["x0"]), n_iter=cs["n_iter"])
            r = dict(x=x.numpy(), If=If.numpy(), n=int(n))
        elif cs["kind"] == "traj":
            dx = LinDx(t(cs["F"]), t(cs["f"]))
            x = util.get_traj(cs["T"], t(cs["u"]), x_init=t(cs["x_init"]), dynamics=dx)
            cost = util.get_cost(cs["T"], t(cs["u"]), QuadCost(t(cs["C"]), t(cs["c"])), dx, x_init=t(cs["x_init"]))
            r = dict(x=x.numpy(), cost=cost.numpy())
        elif cs["kind"] == "env":
            import importlib
            mod = importlib.import_module("mpc.env_dx." + cs["env"])
            if cs["env"] == "pendulum":
                dx = mod.PendulumDx(params=t(cs["params"]), simple=cs["simple"])
            else:
                dx = mod.CartpoleDx(params=t(cs["params"]))
            x, u = t(cs["x"]).requires_grad_(True), t(cs["u"]).requires_grad_(True)
            y = dx(x, u)
            # the Jacobian MPC.linearize_dynamics(AUTO_DIFF) forms (mpc/mpc.py:528-549): one backward per output
            J = []
            for k in range(y.shape[1]):
                gx, gu = torch.autograd.grad(y[:, k].sum(), (x, u), retain_graph=True)
                J.append(torch.cat((gx, gu), 1))
            r = dict(y=y.detach().numpy(), J=torch.stack(J, 1).numpy())
        else:
            net = NNDynamics(cs["ns"], cs["nc"], hidden_sizes=list(cs["hidden"]), activation=cs["act"], passthrough=cs["passthrough"]).double()
            with torch.no_grad():
                for fc, W, b in zip(net.fcs, cs["Ws"], cs["bs"]):
                    fc.weight.copy_(t(W)); fc.bias.copy_(t(b))
            x, u = t(cs["x"]), t(cs["u"])
            y = net(x, u)
            R, S = net.grad_input(x, u)
            r = dict(y=y.detach().numpy(), J=torch.cat((R, S), 2).detach().numpy())
    except Exception as e:
        r = dict(error=type(e).__name__ + ": " + str(e)[:200])
    out.append(r)
pickle.dump(out, open(sys.argv[2], "wb"))
