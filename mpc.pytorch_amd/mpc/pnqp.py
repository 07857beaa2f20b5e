"""`pnqp(H, q, lower, upper, x_init=None, n_iter=20)` -- batched projected-Newton box QP.

Same call and return convention as the reference's mpc/pnqp.py:5-82; the solve itself is the
`mpc_pnqp` kernel (one workgroup per problem, the working matrix in LDS).  Each problem iterates
on its own (the reference's loops are batch-global; per-problem == the reference at n_batch=1).
"""
import torch

from . import _native


def pnqp(H, q, lower, upper, x_init=None, n_iter=20):
    """min_x 0.5 x'Hx + q'x  s.t. lower <= x <= upper, batched over dim 0.

    Returns (x, H_factor, If, n_iter) as the reference does (:59, :82): `H_factor` is the free-set
    Hessian H_ itself for n == 1 and `(LU, pivots)` of H_ (usable with torch.lu_solve) otherwise;
    `If` is the free-set indicator as a float tensor; `n_iter` the largest per-problem iteration
    index."""
    n_batch, n, _ = H.size()
    r = _native.backend().pnqp(H, q, lower, upper, x_init=x_init, n_iter=n_iter)
    its = r["iters"]
    unconverged = r["status"]
    n_it, bad = (int(v) for v in torch.stack((its.max(), unconverged.max())).tolist())
    if bad:
        print("[WARNING] pnqp warning: Did not converge")      # reference :81
    If = r["If"].to(H.dtype)
    if n == 1:
        fac = r["Hfree"]
    else:
        fac = tuple(torch.linalg.lu_factor(r["Hfree"]))
    return r["x"], fac, If, n_it
