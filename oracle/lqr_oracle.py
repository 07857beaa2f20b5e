"""ctypes front-end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package (mpc.pytorch_amd/) never does.  The oracle
restates mpc/lqr_step.py, mpc/pnqp.py and mpc/util.py of locuslab/mpc.pytorch
in C (oracle/lqr_oracle_impl.inc, each function citing the reference lines it
follows) and is pinned against outputs of the unmodified reference
(tests/golden/, tests/test_oracle_golden.py).
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Cfg(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int), ("T", ctypes.c_int), ("ns", ctypes.c_int), ("nc", ctypes.c_int),
                ("bound_mode", ctypes.c_int), ("lo_s", ctypes.c_double), ("hi_s", ctypes.c_double),
                ("delta_u", ctypes.c_double), ("ls_decay", ctypes.c_double),
                ("max_ls_iter", ctypes.c_int), ("pnqp_iter", ctypes.c_int),
                ("lockstep", ctypes.c_int), ("nthreads", ctypes.c_int), ("qp_cold", ctypes.c_int)]


def build(force=False):
    so = os.path.join(_HERE, "liblqr_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("lqr_oracle.c", "lqr_oracle_impl.inc", "lqr_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liblqr_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def max_threads():
    return int(lib().lqr_oracle_max_threads())


def _sfx(dtype):
    return {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[np.dtype(dtype)]


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _arr(a, dtype):
    return None if a is None else np.ascontiguousarray(np.asarray(a), dtype=dtype)


def _bounds(u_lower, u_upper, T, B, nc, dtype):
    """float or [T,B,nc] array bounds -> (mode, lo_s, hi_s, lo, hi)"""
    if u_lower is None:
        return 0, 0.0, 0.0, None, None
    if isinstance(u_lower, float) and isinstance(u_upper, float):
        return 1, u_lower, u_upper, None, None
    lo = np.broadcast_to(np.asarray(u_lower, dtype=dtype), (T, B, nc))
    hi = np.broadcast_to(np.asarray(u_upper, dtype=dtype), (T, B, nc))
    return 2, 0.0, 0.0, np.ascontiguousarray(lo), np.ascontiguousarray(hi)


def lqr_step(x_init, C, c, F, f, cur_x, cur_u, u_lower=None, u_upper=None, u_zero_I=None,
             delta_u=None, linesearch_decay=0.2, max_linesearch_iter=10, lockstep=False,
             nthreads=1, pnqp_iter=20, return_gains=False, qp_cold=False):
    """LQRStepFn.forward (mpc/lqr_step.py:277-309) on numpy arrays.

    lockstep=True  -> the reference called with the whole batch;
    lockstep=False -> the reference called once per problem (n_batch = 1).
    qp_cold=True   -> every box QP whose Quu is positive definite takes pnqp's own cold start (mpc/pnqp.py:14-19) instead of
                      k of timestep t+1 (mpc/lqr_step.py:137,141): lqr_oracle.h, the start of the fused float32 kernels.
    Returns dict(new_x, new_u, costs, old_costs, full_du_norm, alpha_du_norm, alphas, n_qp_iter[, K, k]).
    """
    C = np.asarray(C)
    dtype = C.dtype
    T, B, n, _ = C.shape
    ns = np.asarray(x_init).shape[1]
    nc = n - ns
    C = _arr(C, dtype); c = _arr(c, dtype); x_init = _arr(x_init, dtype)
    F = _arr(F, dtype)
    f = None if (f is None or np.asarray(f).size == 0) else _arr(f, dtype)
    cur_x = _arr(cur_x, dtype); cur_u = _arr(cur_u, dtype)
    mode, lo_s, hi_s, lo, hi = _bounds(u_lower, u_upper, T, B, nc, dtype)
    zm = None if u_zero_I is None else np.ascontiguousarray(np.asarray(u_zero_I).astype(np.uint8))
    cfg = _Cfg(B, T, ns, nc, mode, lo_s, hi_s, float("nan") if delta_u is None else float(delta_u),
               float(linesearch_decay), int(max_linesearch_iter), int(pnqp_iter), int(bool(lockstep)),
               int(nthreads), int(bool(qp_cold)))
    out = dict(new_x=np.empty((T, B, ns), dtype), new_u=np.empty((T, B, nc), dtype),
               costs=np.empty(B, dtype), old_costs=np.empty(B, dtype), full_du_norm=np.empty(B, dtype),
               alpha_du_norm=np.empty(B, dtype), alphas=np.empty(B, dtype))
    K = np.empty((T, B, nc, ns), dtype) if return_gains else None
    k = np.empty((T, B, nc), dtype) if return_gains else None
    nq = ctypes.c_int(0)
    fn = getattr(lib(), "lqr_oracle_step_" + _sfx(dtype))
    fn.restype = ctypes.c_int
    rc = fn(ctypes.byref(cfg), _p(x_init), _p(C), _p(c), _p(F), _p(f), _p(cur_x), _p(cur_u), _p(lo), _p(hi),
            _p(zm), _p(out["new_x"]), _p(out["new_u"]), _p(out["costs"]), _p(out["old_costs"]),
            _p(out["full_du_norm"]), _p(out["alpha_du_norm"]), _p(out["alphas"]), ctypes.byref(nq), _p(K), _p(k))
    assert rc == 0
    out["n_qp_iter"] = nq.value
    if return_gains:
        out["K"], out["k"] = K, k
    return out


def pnqp(H, q, lower, upper, x_init=None, n_iter=20, lockstep=False):
    """mpc/pnqp.py:5-82.  Returns dict(x, If, iters, converged, Hfac, piv)."""
    H = np.asarray(H)
    dtype = H.dtype
    B, n, _ = H.shape
    H = _arr(H, dtype); q = _arr(q, dtype)
    lo = np.ascontiguousarray(np.broadcast_to(np.asarray(lower, dtype=dtype), (B, n)))
    hi = np.ascontiguousarray(np.broadcast_to(np.asarray(upper, dtype=dtype), (B, n)))
    x0 = _arr(x_init, dtype)
    x = np.empty((B, n), dtype); Hf = np.empty((B, n, n), dtype)
    piv = np.empty((B, n), np.int32); If = np.empty((B, n), np.uint8)
    iters = np.empty(B, np.int32); conv = np.empty(B, np.int32)
    fn = getattr(lib(), "lqr_oracle_pnqp_" + _sfx(dtype))
    rc = fn(B, n, int(bool(lockstep)), int(n_iter), _p(H), _p(q), _p(lo), _p(hi), _p(x0),
            _p(x), _p(Hf), _p(piv), _p(If), _p(iters), _p(conv))
    assert rc == 0
    return dict(x=x, If=If, iters=iters, converged=conv, Hfac=Hf, piv=piv)


def traj_cost(x_init, u, F, f=None, C=None, c=None):
    """util.get_traj (LinDx) + util.get_cost (QuadCost), mpc/util.py:102-153."""
    u = np.asarray(u)
    dtype = u.dtype
    T, B, nc = u.shape
    ns = np.asarray(x_init).shape[1]
    x_init = _arr(x_init, dtype); u = _arr(u, dtype); F = _arr(F, dtype)
    f = None if (f is None or np.asarray(f).size == 0) else _arr(f, dtype)
    C = _arr(C, dtype); c = _arr(c, dtype)
    x = np.empty((T, B, ns), dtype)
    cost = np.empty(B, dtype) if C is not None else None
    fn = getattr(lib(), "lqr_oracle_traj_cost_" + _sfx(dtype))
    rc = fn(B, T, ns, nc, _p(x_init), _p(u), _p(C), _p(c), _p(F), _p(f), _p(x), _p(cost))
    assert rc == 0
    return x, cost


def kkt_backward(C, c, F, f, new_x, new_u, dl_dx, dl_du, u_lower=None, u_upper=None,
                 lockstep=False, nthreads=1):
    """LQRStepFn.backward (mpc/lqr_step.py:312-407).

    Returns dict(dx_init, dC, dc, dF, df, dx, du).  df is None when f is empty.
    """
    C = np.asarray(C)
    dtype = C.dtype
    T, B, n, _ = C.shape
    ns = np.asarray(new_x).shape[2]
    nc = n - ns
    C = _arr(C, dtype); c = _arr(c, dtype); F = _arr(F, dtype)
    has_f = not (f is None or np.asarray(f).size == 0)
    new_x = _arr(new_x, dtype); new_u = _arr(new_u, dtype)
    dl_dx = _arr(dl_dx, dtype); dl_du = _arr(dl_du, dtype)
    mode, lo_s, hi_s, lo, hi = _bounds(u_lower, u_upper, T, B, nc, dtype)
    cfg = _Cfg(B, T, ns, nc, mode, lo_s, hi_s, float("nan"), 0.2, 10, 20, int(bool(lockstep)), int(nthreads))
    out = dict(dx_init=np.empty((B, ns), dtype), dC=np.empty((T, B, n, n), dtype), dc=np.empty((T, B, n), dtype),
               dF=np.zeros(F.shape, dtype), df=np.empty((T - 1, B, ns), dtype) if has_f else None,
               dx=np.empty((T, B, ns), dtype), du=np.empty((T, B, nc), dtype))
    fn = getattr(lib(), "lqr_oracle_kkt_backward_" + _sfx(dtype))
    rc = fn(ctypes.byref(cfg), _p(C), _p(c), _p(F), int(has_f), _p(new_x), _p(new_u), _p(lo), _p(hi),
            _p(dl_dx), _p(dl_du), _p(out["dx_init"]), _p(out["dC"]), _p(out["dc"]), _p(out["dF"]),
            _p(out["df"]), _p(out["dx"]), _p(out["du"]))
    assert rc == 0
    return out
