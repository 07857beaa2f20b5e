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
    return r["x"], fac, If, _LazyIterCount(r["iters"], r["status"])


class _LazyIterCount:
    """The 4th return value of pnqp (the reference's `i`, mpc/pnqp.py:59, 82): the largest per-problem iteration
    index.  Computing it needs a device->host read, and so does the reference's "Did not converge" warning (:81);
    both happen the first time the number is LOOKED AT (int(), comparison, arithmetic, formatting, range()), not
    inside pnqp() -- a solve whose caller only wants x stays asynchronous.  Quacks like the int it stands for
    (__int__ / __index__ and the arithmetic / comparison protocol) without being one."""

    def __init__(self, iters, status):
        self._dev, self._val = (iters, status), None

    def _get(self):
        if self._val is None:
            its, bad = self._dev
            n_it, unconverged = (int(v) for v in torch.stack((its.max(), bad.max())).tolist())
            if unconverged:
                print("[WARNING] pnqp warning: Did not converge")      # reference :81
            self._val, self._dev = n_it, None
        return self._val

    def __int__(self): return self._get()
    def __index__(self): return self._get()
    def __float__(self): return float(self._get())
    def __repr__(self): return repr(self._get())
    def __str__(self): return str(self._get())
    def __format__(self, spec): return format(self._get(), spec)
    def __hash__(self): return hash(self._get())
    def __bool__(self): return bool(self._get())
    def __eq__(self, o): return self._get() == o
    def __ne__(self, o): return self._get() != o
    def __lt__(self, o): return self._get() < o
    def __le__(self, o): return self._get() <= o
    def __gt__(self, o): return self._get() > o
    def __ge__(self, o): return self._get() >= o
    def __add__(self, o): return self._get() + o
    def __radd__(self, o): return o + self._get()
    def __sub__(self, o): return self._get() - o
    def __rsub__(self, o): return o - self._get()
    def __mul__(self, o): return self._get() * o
    def __rmul__(self, o): return o * self._get()
    def __neg__(self): return -self._get()
    def __pos__(self): return +self._get()
    def __abs__(self): return abs(self._get())
    def __truediv__(self, o): return self._get() / o
    def __rtruediv__(self, o): return o / self._get()
    def __floordiv__(self, o): return self._get() // o
    def __rfloordiv__(self, o): return o // self._get()
    def __mod__(self, o): return self._get() % o
    def __rmod__(self, o): return o % self._get()
    def __pow__(self, o): return self._get() ** o
    def __rpow__(self, o): return o ** self._get()
    def __divmod__(self, o): return divmod(self._get(), o)
    def __round__(self, n=None): return round(self._get(), n) if n is not None else self._get()

    def __getattr__(self, name):           # anything else an int has (bit_length, to_bytes, real, ...): the int's own
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._get(), name)
