"""Pins the CPU oracle (oracle/) against outputs of the UNMODIFIED reference.

The fixtures under tests/golden/ were produced by tests/golden/make_golden.py,
which imports locuslab/mpc.pytorch from /root/reference in the build container.
If these tests pass the oracle is a faithful restatement of
mpc/lqr_step.py, mpc/pnqp.py and mpc/util.py -- "parity pinned".
"""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden
from helpers import bounds_of, step_kwargs, close_with_ref_noise, scrambled_du_norm
from oracle import lqr_oracle as O

STEP_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "step_*.npz")))
GRAD_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "grad_*.npz")))
PNQP_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "pnqp_*.npz")))


def test_fixture_inventory():
    assert len(STEP_CASES) >= 17 and len(GRAD_CASES) >= 6 and len(PNQP_CASES) >= 5


@pytest.mark.parametrize("name", STEP_CASES)
@pytest.mark.parametrize("mode", ["batch", "pp"])
def test_lqr_step_matches_reference(name, mode):
    """LQRStepFn.forward (mpc/lqr_step.py:277-309), whole batch and per problem."""
    z = golden(name)
    lock = mode == "batch"
    o = O.lqr_step(lockstep=lock, **step_kwargs(z))
    f64 = z["C"].dtype == np.float64
    if f64:
        tol = dict(rtol=1e-9, atol=1e-9)
        if "singular_cfg5" in name:
            # an 8x8 pseudo-inverse through two different SVDs (LAPACK's in the reference, one-sided Jacobi here)
            tol = dict(rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(o["new_x"], z["new_x_" + mode], **tol)
        np.testing.assert_allclose(o["new_u"], z["new_u_" + mode], **tol)
        np.testing.assert_allclose(o["costs"], z["costs_" + mode], **tol)
    else:
        # fp32: rtol 1e-3 / atol 1e-4, widened only by the reference's OWN
        # fp32 deviation from its float64 run on the same inputs (pnqp stops at |dx| < 1e-4,
        # so an fp32 run of the reference is itself only reproducible to that level).
        noise_x = np.abs(z["new_x_pp"] - z["new_x_ref64"])
        noise_u = np.abs(z["new_u_pp"] - z["new_u_ref64"])
        close_with_ref_noise(o["new_x"], z["new_x_" + mode], noise_x, 1e-3, 1e-4)
        close_with_ref_noise(o["new_u"], z["new_u_" + mode], noise_u, 1e-3, 1e-4)
        # (cost: the same widening -- on step_singular_cfg5_f32 the reference's float32 8x8 pinverse is noise along the
        #  dead control and its own float32 cost is 2.3x its float64 cost)
        close_with_ref_noise(o["costs"], z["costs_" + mode], np.abs(z["costs_pp"] - z["costs_ref64"]), 1e-4, 0.0)
        # and against the reference's float64 run directly (a per-problem run: with a non-symmetric Quu the box QP
        # of the reference needs 2x the trips, and its batch-global loop then leaves the coupled problems elsewhere
        # than their own solves -- 3e-2 on step_asym_ns_bounded_f32 -- so the batch flavour is held to the batch run only)
        if not (lock and "asym" in name and "u_lower" in z):
            np.testing.assert_allclose(o["new_x"], z["new_x_ref64"], rtol=1e-3, atol=1e-4)
            np.testing.assert_allclose(o["new_u"], z["new_u_ref64"], rtol=1e-3, atol=1e-4)
    # line-search step sizes
    if lock:
        np.testing.assert_allclose(o["alphas"].mean(), z["mean_alphas_batch"][0], rtol=1e-6)
        if f64:
            assert o["n_qp_iter"] == int(z["n_qp_batch"][0])
    else:
        np.testing.assert_allclose(o["alphas"], z["alphas_pp"], rtol=1e-6)
        if f64:
            assert o["n_qp_iter"] == int(z["n_qp_pp"].max())


@pytest.mark.parametrize("name", STEP_CASES)
def test_full_du_norm_and_reference_scramble(name):
    """The per-problem ||u - u'||_2 equals the reference at n_batch = 1; for n_batch > 1 the
    reference's `.transpose(1,2).contiguous().view(n_batch,-1)` (mpc/lqr_step.py:243-245)
    mixes problems -- documented quirk, reproduced here from the oracle's own du."""
    z = golden(name)
    if name == "step_masked_nc1_f64":
        pytest.skip("per-problem fixture of this case was produced from duplicated pairs")
    tol = 1e-9 if z["C"].dtype == np.float64 else 2e-3
    o = O.lqr_step(lockstep=False, **step_kwargs(z))
    np.testing.assert_allclose(o["full_du_norm"], z["full_du_norm_pp"], rtol=tol, atol=tol)
    ob = O.lqr_step(lockstep=True, **step_kwargs(z))
    # alpha = 1 on the first pass: recompute that pass's du from a 1-pass line search
    kw = step_kwargs(z)
    kw["max_linesearch_iter"] = 1
    o1 = O.lqr_step(lockstep=True, **kw)
    scr = scrambled_du_norm(z["cur_u"] - o1["new_u"])
    np.testing.assert_allclose(scr, z["full_du_norm_batch"], rtol=tol, atol=tol)
    np.testing.assert_allclose(np.sqrt((ob["full_du_norm"] ** 2).sum()),
                               np.sqrt((z["full_du_norm_batch"].astype(np.float64) ** 2).sum()), rtol=max(tol, 1e-6))


@pytest.mark.parametrize("name", GRAD_CASES)
@pytest.mark.parametrize("lock", [True, False])
def test_kkt_backward_matches_reference(name, lock):
    """LQRStepFn.backward (mpc/lqr_step.py:312-407): dx_init, dC, dc, dF, df."""
    z = golden(name)
    beta = float(z["beta"][0])
    lo, hi = (None, None) if np.isnan(beta) else (-beta, beta)
    o = O.kkt_backward(z["C"], z["c"], z["F"], z.get("f"), z["x"], z["u"], z["dl_dx"], z["dl_du"], lo, hi,
                       lockstep=lock)
    f64 = z["C"].dtype == np.float64
    for k in ("dx_init", "dC", "dc", "dF", "df"):
        if k not in z:
            assert o[k] is None
            continue
        scale = max(1.0, np.abs(z[k]).max())
        np.testing.assert_allclose(o[k] / scale, z[k] / scale, rtol=0, atol=1e-10 if f64 else 2e-5)


@pytest.mark.parametrize("name", ["jac_unconstrained", "jac_constrained"])
def test_kkt_backward_full_jacobians(name):
    """tests/test_mpc.py:303-395 / :398-500 inputs: every row of du/d{x_init,C,c,F,f}."""
    z = golden(name)
    ns, nc, T, B = (int(v) for v in z["meta"])
    beta = float(z["beta"][0])
    if name == "jac_constrained":   # the reference test's own precondition (:453-454)
        at = np.isclose(np.abs(z["u"]), beta, atol=1e-8)
        assert at.any() and not at.all()
    for i in range(T * B * nc):
        g = np.zeros((T, B, nc))
        g.reshape(-1)[i] = 1.0
        o = O.kkt_backward(z["C"], z["c"], z["F"], z["f"], z["x"], z["u"], np.zeros((T, B, ns)), g, -beta, beta)
        for k in ("dx_init", "dC", "dc", "dF", "df"):
            np.testing.assert_allclose(o[k].reshape(-1), z["J_" + k][i], rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("name", PNQP_CASES)
@pytest.mark.parametrize("mode", ["batch", "pp"])
def test_pnqp_matches_reference(name, mode):
    """mpc/pnqp.py:5-82 incl. tests/test_mpc.py:65-88's n=100 problem."""
    z = golden(name)
    o = O.pnqp(z["H"], z["q"], z["lower"], z["upper"], z.get("x0"), lockstep=(mode == "batch"))
    f64 = z["H"].dtype == np.float64
    np.testing.assert_allclose(o["x"], z["x_" + mode], rtol=0, atol=1e-10 if f64 else 5e-6)
    assert np.array_equal(o["If"], z["If_" + mode].astype(np.uint8))
    if f64:
        ref_it = z["iters_" + mode]
        assert o["iters"].tolist() == (ref_it.tolist() * len(o["iters"]) if mode == "batch" else ref_it.tolist())
    assert o["converged"].all()
    # box feasibility
    assert (o["x"] >= z["lower"]).all() and (o["x"] <= z["upper"]).all()


def test_traj_and_cost_match_reference():
    """util.get_traj / util.get_cost, mpc/util.py:102-153."""
    z = golden("traj_cost")
    x, cost = O.traj_cost(z["x_init"], z["u"], z["F"], z["f"], z["C"], z["c"])
    np.testing.assert_allclose(x, z["x"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(cost, z["cost"], rtol=1e-12)


def test_oracle_edge_cases():
    """T = 1 (no dynamics step at all), B = 1, and the empty-f path."""
    rng = np.random.default_rng(0)
    ns, nc, n = 2, 2, 4
    A = rng.standard_normal((1, 1, n, n))
    C = A.transpose(0, 1, 3, 2) @ A + np.eye(n)
    c = rng.standard_normal((1, 1, n))
    F = np.zeros((0, 1, ns, n))
    x0 = rng.standard_normal((1, ns))
    o = O.lqr_step(x0, C, c, F, None, x0[None], np.zeros((1, 1, nc)))
    # single stage: u* = -Cuu^{-1} (Cux x0 + c_u)
    u_star = -np.linalg.solve(C[0, 0, ns:, ns:], C[0, 0, ns:, :ns] @ x0[0] + c[0, 0, ns:])
    np.testing.assert_allclose(o["new_u"][0, 0], u_star, rtol=1e-10)
    np.testing.assert_allclose(o["new_x"][0], x0)


# ---------------------------------------------------------------------------------------------
# oracle/env_oracle.py: the shipped simulators, pinned on the reference's own modules
# ---------------------------------------------------------------------------------------------
ENV_CASES = [("env_pendulum_f64", 1), ("env_pendulum_full_f64", 2), ("env_cartpole_f64", 3)]


@pytest.mark.parametrize("name,kind", ENV_CASES)
def test_env_oracle_matches_reference_modules(name, kind):
    """step == PendulumDx / CartpoleDx.forward; linearize == MPC.linearize_dynamics(AUTO_DIFF), which
    re-rolls the trajectory from x[0] (mpc/mpc.py:524-527, 592-593); rollout == LQRStep with the module
    as true_dynamics (mpc/lqr_step.py:223-225)."""
    from oracle import env_oracle as E
    z = golden(name)
    ns, nc, T, B = (int(v) for v in z["meta"])
    prm = z["params"]
    x, u = z["x"][:-1].reshape(-1, ns), z["u"][:-1].reshape(-1, nc)
    np.testing.assert_allclose(E.step(kind, x, u, prm), z["next"].reshape(-1, ns), rtol=1e-13, atol=1e-13)
    xr = E.traj(kind, z["x"][0], z["u"], prm)
    F, f = E.linearize(kind, xr[:-1].reshape(-1, ns), u, prm)
    np.testing.assert_allclose(F, z["F"].reshape(F.shape), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(f, z["f"].reshape(f.shape), rtol=1e-9, atol=1e-9)
    # controls past the module's clamp are in the pendulum fixtures: their column of F is exactly zero
    sat = np.abs(u[:, 0]) > E.u_max_of(kind)
    assert (kind == 3 or sat.any()) and np.all(F[sat][:, :, ns] == 0)
    # one LQR step: sweep (C oracle) + rollout through the simulator
    o = O.lqr_step(z["x_init"], z["Q"], z["p"], z["step_F"], z["step_f"], z["step_cur_x"], z["step_cur_u"],
                   float(z["lower"][0]), float(z["upper"][0]), linesearch_decay=float(z["decay"][0]),
                   max_linesearch_iter=int(z["max_ls"][0]), return_gains=True)
    nx, nu, costs, full, alphas = E.rollout(kind, prm, z["x_init"], z["Q"], z["p"], o["K"], o["k"], z["step_cur_x"],
                                            z["step_cur_u"], float(z["lower"][0]), float(z["upper"][0]),
                                            float(z["decay"][0]), int(z["max_ls"][0]))
    np.testing.assert_allclose(nx, z["step_new_x"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(nu, z["step_new_u"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(costs, z["step_costs"], rtol=1e-9)
    # the batched form the full-batch GPU tests use: the same per-problem line search, trial by trial
    bx, bu, bc, bfull, balpha, trials, old = E.rollout_batched(
        kind, prm, z["x_init"], z["Q"], z["p"], o["K"], o["k"], z["step_cur_x"], z["step_cur_u"],
        float(z["lower"][0]), float(z["upper"][0]), float(z["decay"][0]), int(z["max_ls"][0]))
    for a, b in ((bx, nx), (bu, nu), (bc, costs), (bfull, full), (balpha, alphas)):
        np.testing.assert_allclose(a, b, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(old, E.quad_cost(z["Q"], z["p"], z["step_cur_x"], z["step_cur_u"]), rtol=1e-13)


NN_CASES = ["nn_sigmoid_f64", "nn_relu2_f64", "nn_headline_f64", "nn_nopass_f64"]


@pytest.mark.parametrize("name", NN_CASES)
def test_mlp_oracle_matches_reference_nndynamics(name):
    """mlp_step == NNDynamics.forward, mlp_jacobian == NNDynamics.grad_input (mpc/dynamics.py:57-128), linearize ==
    MPC.linearize_dynamics(ANALYTIC) (mpc/mpc.py:495-512), rollout == LQRStep with the module as true_dynamics."""
    from oracle import env_oracle as E
    z = golden(name)
    ns, nc, T, B = (int(v) for v in z["meta"][:4])
    net = E.Mlp.from_npz(z)
    np.testing.assert_allclose(E.mlp_step(z["px"], z["pu"], net), z["pnext"], rtol=1e-12, atol=1e-12)
    J = E.mlp_jacobian(z["px"], z["pu"], net)
    np.testing.assert_allclose(J[:, :, :ns], z["pR"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(J[:, :, ns:], z["pS"], rtol=1e-12, atol=1e-12)
    if net.activation == "sigmoid":            # smooth: the closed form is also the derivative of mlp_step
        h = 1e-5
        for j in range(ns + nc):
            e = np.zeros(ns + nc)
            e[j] = h
            d = (E.mlp_step(z["px"] + e[:ns], z["pu"] + e[ns:], net) - E.mlp_step(z["px"] - e[:ns], z["pu"] - e[ns:], net)) / (2 * h)
            np.testing.assert_allclose(J[:, :, j], d, rtol=1e-6, atol=1e-8)
    xr = E.traj(E.MLP, z["x_init"], z["step_cur_u"], net)
    np.testing.assert_allclose(xr, z["step_cur_x"], rtol=1e-12, atol=1e-12)
    F, f = E.linearize(E.MLP, xr[:-1].reshape(-1, ns), z["step_cur_u"][:-1].reshape(-1, nc), net)
    np.testing.assert_allclose(F, z["step_F"].reshape(F.shape), rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(f, z["step_f"].reshape(f.shape), rtol=1e-11, atol=1e-11)
    bound = float(z["bound"][0])
    lo, hi = (None, None) if np.isnan(bound) else (-bound, bound)
    o = O.lqr_step(z["x_init"], z["C"], z["c"], z["step_F"], z["step_f"], z["step_cur_x"], z["step_cur_u"], lo, hi,
                   linesearch_decay=float(z["decay"][0]), max_linesearch_iter=int(z["max_ls"][0]), return_gains=True)
    nx, nu, costs, full, alphas = E.rollout(E.MLP, net, z["x_init"], z["C"], z["c"], o["K"], o["k"], z["step_cur_x"],
                                            z["step_cur_u"], lo, hi, float(z["decay"][0]), int(z["max_ls"][0]))
    np.testing.assert_allclose(nx, z["step_new_x"], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(nu, z["step_new_u"], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(costs, z["step_costs"], rtol=1e-8)
    # (full_du_norm of a whole-batch reference call is not comparable: mpc/lqr_step.py:243-245 views the transposed
    # [T, nc, B] difference as [B, T nc], which mixes the problems unless n_batch = 1)
    bx, bu, bc, bfull, balpha, _, _ = E.rollout_batched(
        E.MLP, net, z["x_init"], z["C"], z["c"], o["K"], o["k"], z["step_cur_x"], z["step_cur_u"], lo, hi,
        float(z["decay"][0]), int(z["max_ls"][0]))
    for a, b in ((bx, nx), (bu, nu), (bc, costs), (bfull, full), (balpha, alphas)):
        np.testing.assert_allclose(a, b, rtol=1e-13, atol=1e-13)


def test_env_oracle_clamp_derivative_on_the_bound():
    """torch.clamp passes the gradient on the CLOSED interval; a control sitting exactly on its bound
    (where a bounded solve puts it) keeps its full derivative."""
    from oracle import env_oracle as E
    x = np.array([[0.3, 0.9, 0.1]])
    F_in, _ = E.linearize(1, x, np.array([[1.0]]), np.array([10., 1., 1.]))
    F_on, _ = E.linearize(1, x, np.array([[2.0]]), np.array([10., 1., 1.]))
    F_out, _ = E.linearize(1, x, np.array([[2.0 + 1e-9]]), np.array([10., 1., 1.]))
    np.testing.assert_allclose(F_on[0, 2, 3], F_in[0, 2, 3], rtol=1e-9)
    assert np.all(F_out[0, :, 3] == 0)
