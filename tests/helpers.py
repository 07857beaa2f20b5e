"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np


def bounds_of(z):
    """(u_lower, u_upper) of a golden fixture: None, python floats, or [T,B,nc] arrays."""
    if "u_lower" not in z:
        return None, None
    lo, hi = z["u_lower"], z["u_upper"]
    if lo.shape == (1,):
        return float(lo[0]), float(hi[0])
    return lo, hi


def step_kwargs(z):
    """Keyword arguments of oracle.lqr_oracle.lqr_step for a step_* fixture."""
    lo, hi = bounds_of(z)
    du = None if np.isnan(z["delta_u"][0]) else float(z["delta_u"][0])
    return dict(x_init=z["x_init"], C=z["C"], c=z["c"], F=z["F"], f=z.get("f"), cur_x=z["cur_x"],
                cur_u=z["cur_u"], u_lower=lo, u_upper=hi, u_zero_I=z.get("u_zero_I"), delta_u=du,
                linesearch_decay=float(z["decay"][0]), max_linesearch_iter=int(z["meta"][5]))


def close_with_ref_noise(actual, desired, ref_noise, rtol, atol):
    """|actual - desired| <= atol + rtol*|desired| + 2*ref_noise, element-wise.

    ref_noise is the reference's own fp32-vs-fp64 deviation on that element; it widens the
    stated tolerance only where the reference cannot reproduce itself any better."""
    err = np.abs(np.asarray(actual, np.float64) - np.asarray(desired, np.float64))
    lim = atol + rtol * np.abs(desired) + 2.0 * ref_noise
    bad = err > lim
    assert not bad.any(), "max excess %.3e at %s (err %.3e, lim %.3e)" % (
        (err - lim).max(), np.unravel_index((err - lim).argmax(), err.shape), err.max(), lim.max())


def scrambled_du_norm(du):
    """The reference's full_du_norm for n_batch > 1: (u-new_u).transpose(1,2).contiguous()
    .view(n_batch,-1).norm(2,1) (mpc/lqr_step.py:243-245) -- a reshape that mixes problems."""
    T, B, nc = du.shape
    return np.sqrt((np.ascontiguousarray(du.transpose(0, 2, 1)).reshape(B, -1) ** 2).sum(1))


def asymmetric_problems(z):
    """[B] bool: the problems of a fixture whose C is not symmetric by the kernels' own test
    (max |C - C'| > 1e-5 max |C| over the horizon: MPC_ST_C_ASYMMETRIC, include/mpc_lqr.h)."""
    C = np.asarray(z["C"], np.float64)
    d = np.abs(C - C.transpose(0, 1, 3, 2)).max(axis=(0, 2, 3))
    return d > 1e-5 * np.abs(C).max(axis=(0, 2, 3))


_PER_PROBLEM = ("new_x", "new_u", "costs", "old_costs", "full_du_norm", "alpha_du_norm", "alphas", "K", "k", "status",
                "qp_iters", "n_qp_pp")


def keep_problems(d, keep):
    """The per-problem outputs of a result dict / golden fixture restricted to the problems in `keep` [B] bool
    (axis 1 of the [T, B, ...] trajectories and gains, axis 0 of the [B] scalars).  Everything else passes through."""
    out = {}
    B = len(keep)
    for k, v in d.items():
        if isinstance(v, np.ndarray) and any(k == n or k.startswith(n + "_") for n in _PER_PROBLEM):
            if v.ndim == 1 and v.shape[0] == B:
                v = v[keep]
            elif v.ndim > 1 and v.shape[1] == B:
                v = v[:, keep]
        out[k] = v
    return out


def check_tie_problems(r, z, rtol=1e-3, atol=1e-4):
    """tests/golden/ties_tight_f32.npz: problems on which the reference's own float32 and float64 runs end on different
    active sets (a QP minimiser on its bound to within rounding).  A kernel result must be ONE of the two branches the
    reference takes, whole trajectory, per problem.  Returns which branch each problem took ("f32" / "f64" / "both")."""
    took = []
    for b in range(z["new_u_pp"].shape[1]):
        ok = {}
        for name, sfx in (("f32", "_pp"), ("f64", "_ref64")):
            eu = np.abs(np.asarray(r["new_u"][:, b], np.float64) - z["new_u" + sfx][:, b])
            ex = np.abs(np.asarray(r["new_x"][:, b], np.float64) - z["new_x" + sfx][:, b])
            ok[name] = bool((eu <= atol + rtol * np.abs(z["new_u" + sfx][:, b])).all() and
                            (ex <= atol + rtol * np.abs(z["new_x" + sfx][:, b])).all() and
                            abs(float(r["costs"][b]) - float(z["costs" + sfx][b])) <= 5e-4 * abs(float(z["costs" + sfx][b])))
        assert ok["f32"] or ok["f64"], "problem %d (%d of the full batch) follows neither branch of the reference" % (b, int(z["which"][b]))
        took.append("both" if ok["f32"] and ok["f64"] else ("f32" if ok["f32"] else "f64"))
    return took
