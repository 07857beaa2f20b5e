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
    r = _native.backend().pnqp(H, q, lower, upper, x_init=x_init, n_iter=n_iter, want_Hfree=(n == 1), want_lu=(n > 1))
    If = r["If"].to(H.dtype)
    # `H_factor`: H_ itself for n == 1 (:50-51), else the (LU, pivots) pair of the last Newton system, as the kernel
    # factorised it (no second factorisation, no rocSOLVER launch)
    fac = r["Hfree"] if n == 1 else (r["LU"], r["pivots"])
    return r["x"], fac, If, _iteration_count(r["iters"], r["status"])


def _iteration_count(iters, status):
    """The 4th return value of pnqp (the reference's `i`, mpc/pnqp.py:59, 82): the largest per-problem iteration index, as
    the 1-element CPU tensor `LQRStep` hands back for n_total_qp_iter -- it arrives by an asynchronous copy and the first
    look at it (int(), a comparison, arithmetic, printing) waits for that copy alone, so a solve whose caller only wants x
    never synchronises.  The reference's "Did not converge" warning (:81) needs the same host read: it is printed at that
    first look (at once for CPU tensors)."""
    from .lqr_step import _host_scalar_async
    n_it = _host_scalar_async(iters.max().reshape(1))
    bad = _host_scalar_async(status.max().reshape(1))

    def warn():
        if float(bad) != 0.0:
            print("[WARNING] pnqp warning: Did not converge")      # reference :81
    if isinstance(n_it, torch.Tensor) and hasattr(n_it, "_settle") and getattr(n_it, "_event", None) is not None:
        n_it._on_settle = warn
    else:
        warn()
    return n_it
