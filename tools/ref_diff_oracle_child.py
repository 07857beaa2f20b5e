#!/usr/bin/env python3
"""Child of tools/ref_diff_oracle.py: the UNMODIFIED reference's LQRStep (forward, and backward through autograd) on pickled cases."""
import os, pickle, sys, warnings, io, contextlib
sys.path.insert(0, os.environ.get("MPC_REFERENCE_DIR", "/root/reference"))
import numpy as np, torch
from mpc import mpc as _m                  # (mpc.mpc first: the reference's modules import each other)
from mpc.lqr_step import LQRStep
from mpc.mpc import QuadCost, LinDx
warnings.filterwarnings("ignore")
cases = pickle.load(open(sys.argv[1], "rb"))
out = []
for cs in cases:
    t = lambda a: None if a is None else torch.from_numpy(a).clone()
    C, c, F, f, x0 = t(cs["C"]), t(cs["c"]), t(cs["F"]), t(cs["f"]), t(cs["x_init"])
    for v in (C, c, F, x0) + ((f,) if f is not None else ()):
        v.requires_grad_(True)
    kw = {k: (t(v) if isinstance(v, np.ndarray) else v) for k, v in cs["kw"].items()}
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            step = LQRStep(n_state=cs["ns"], n_ctrl=cs["nc"], T=cs["T"], true_cost=QuadCost(C.detach(), c.detach()),
                           true_dynamics=LinDx(F.detach(), None if f is None else f.detach()), delta_space=True,
                           current_x=t(cs["cur_x"]), current_u=t(cs["cur_u"]), **kw)
            x, u, nqp, costs, du, ma = step(x0, C, c, F, f if f is not None else torch.Tensor())
        r = dict(new_x=x.detach().numpy(), new_u=u.detach().numpy(), costs=costs.detach().numpy(), n_qp=float(nqp), du=du.detach().numpy())
        if cs["grads"]:
            # (the reference's backward takes the gradients of TWO outputs: it is reachable through the no-op forward only, the way
            # MPC.forward attaches it at its best iterate, mpc/mpc.py:318-319)
            with contextlib.redirect_stdout(io.StringIO()):
                step2 = LQRStep(n_state=cs["ns"], n_ctrl=cs["nc"], T=cs["T"], true_cost=QuadCost(C.detach(), c.detach()),
                                true_dynamics=LinDx(F.detach(), None if f is None else f.detach()), delta_space=True,
                                current_x=x.detach(), current_u=u.detach(), no_op_forward=True, **kw)
                x2, u2 = step2(x0, C, c, F, f if f is not None else torch.Tensor())
            ((x2 * t(cs["wx"])).sum() + (u2 * t(cs["wu"])).sum()).backward()
            r.update(dC=C.grad.numpy(), dc=c.grad.numpy(), dF=F.grad.numpy(), dx_init=x0.grad.numpy(), df=None if f is None else f.grad.numpy())
    except Exception as e:
        r = dict(error=type(e).__name__ + ": " + str(e)[:200])
    out.append(r)
pickle.dump(out, open(sys.argv[2], "wb"))
