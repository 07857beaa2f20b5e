"""CPU-only tests: the C ABI loads and exports what include/mpc_lqr.h declares, the product refuses
CPU tensors (no fallback), and the HOST logic of the package (MPC.forward's iteration rules, the
autograd wiring, argument handling) reproduces the reference's results when the kernels are
replaced by the oracle (tests/oracle_backend.py -- a test hook, not a product path)."""
import contextlib
import ctypes
import io
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden
from mpc import _native, mpc, util
from mpc.lqr_step import LQRStep
from mpc.mpc import GradMethods, LinDx, QuadCost
from oracle_backend import OracleBackend


@pytest.fixture
def oracle_backend():
    be = OracleBackend()
    prev = _native.set_backend_for_testing(be)
    yield be
    _native.set_backend_for_testing(prev)


def tt(z, k):
    return torch.from_numpy(z[k]) if k in z else None


# ---------------------------------------------------------------------------------------------
# the boundary
# ---------------------------------------------------------------------------------------------
def test_library_loads_and_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mpc_lqr.h")).read()
    declared = set(re.findall(r"\b(mpc_[a-z_]+)\s*\(", header))
    assert {"mpc_lqr_step", "mpc_lqr_sweep", "mpc_lqr_rollout", "mpc_lqr_kkt_grads", "mpc_pnqp",
            "mpc_traj_cost", "mpc_select_best"} <= declared
    L = _native.load()
    for name in declared:
        assert hasattr(L, name), name
    assert set(_native.EXPORTS) == declared
    assert L.mpc_lqr_abi_version() == _native.ABI_VERSION == 9
    assert b"gfx950" in L.mpc_lqr_build_info()


def test_struct_layout_matches_header():
    # sizeof checks guard the ctypes mirror against drift from include/mpc_lqr.h
    assert ctypes.sizeof(_native.Problem) == 6 * 4 + 8 + 4 * (8 + 16) + 16
    assert ctypes.sizeof(_native.Options) == 8 + 16 + 24 + 16 + 8 + 8 + 24      # (+ qp_start and its two strides: ABI 8)
    assert ctypes.sizeof(_native.EnvDynamics) == 8 + 8 + 16
    assert ctypes.sizeof(_native.Outputs) == 11 * 8
    assert ctypes.sizeof(_native.MlpDynamics) == 4 * 4 + 5 * 4 + 4 + 4 * 8 + 4 * 8       # 9 ints + padding to 8


def test_argument_validation_without_gpu():
    L = _native.load()
    p = _native.Problem()
    p.B, p.T, p.ns, p.nc, p.dtype = 4, 0, 3, 1, 0
    o, out = _native.Options(), _native.Outputs()
    o.max_linesearch_iter, o.delta_u = 10, float('nan')
    rc = L.mpc_lqr_step(ctypes.byref(p), ctypes.byref(o), ctypes.byref(out), None, 0, 0, None)
    assert rc == -1 and b"T>=1" in L.mpc_lqr_last_error()
    p.T, p.dtype = 5, 7
    assert L.mpc_lqr_step(ctypes.byref(p), ctypes.byref(o), ctypes.byref(out), None, 0, 0, None) == -3
    p.dtype = 0
    assert L.mpc_lqr_step(ctypes.byref(p), ctypes.byref(o), ctypes.byref(out), None, 0, 0, None) == -2   # NULL C
    p.B = 0      # empty batch is a no-op, not an error
    assert L.mpc_lqr_step(ctypes.byref(p), ctypes.byref(o), ctypes.byref(out), None, 0, 0, None) == 0
    assert L.mpc_pnqp(0, 0, 4, None, None, None, None, None, 20, None, None, None, None, None, None) == 0
    assert L.mpc_pnqp(3, 1, 4, None, None, None, None, None, 20, None, None, None, None, None, None) == -3


def test_kernel_routing_queries_without_gpu():
    """mpc_lqr_impl_supported answers from sizes and dtype alone (no GPU needed): round 6's padded 12/4 instantiation (impl 8) takes every
    float32 shape up to 12/4 -- 12/4 itself included: blocks the exact kernel refuses for their alignment --, nothing beyond, no float64;
    mpc_du_norm_reference validates its arguments."""
    be = _native.HipBackend()
    for ns, nc in ((12, 4), (10, 3), (8, 4), (1, 1), (12, 1), (3, 4)):
        assert be.impl_supported(ns, nc, torch.float32, _native.IMPL_DPP16_PAD), (ns, nc)
        assert not be.impl_supported(ns, nc, torch.float64, _native.IMPL_DPP16_PAD)
    for ns, nc in ((13, 4), (12, 5), (32, 8)):
        assert not be.impl_supported(ns, nc, torch.float32, _native.IMPL_DPP16_PAD)
        assert be.impl_supported(ns, nc, torch.float32, _native.IMPL_MFMA40_PAD)
    assert be.impl_supported(12, 4, torch.float32, _native.IMPL_DPP16) and not be.impl_supported(10, 3, torch.float32, _native.IMPL_DPP16)
    L = _native.load()
    assert L.mpc_du_norm_reference(0, 5, 0, 2, None, None, None, None) == 0           # empty batch: a no-op
    assert L.mpc_du_norm_reference(0, 5, 3, 2, None, None, None, None) == -2          # NULL arrays
    assert L.mpc_du_norm_reference(7, 5, 3, 2, None, None, None, None) == -3          # bad dtype
    assert L.mpc_du_norm_reference(0, 0, 3, 2, None, None, None, None) == -1          # bad dims


def test_network_and_sweep_only_entry_points_validate_arguments_without_gpu():
    """mpc_mlp_* (NNDynamics in the kernels) and MPC_OPT_SWEEP_ONLY reject bad arguments before any launch."""
    L = _native.load()
    net = _native.MlpDynamics()
    assert L.mpc_mlp_workspace_bytes(ctypes.byref(net)) == 0                      # no layers
    net.n_layers, net.activation, net.passthrough = 2, 0, 1
    net.widths[0], net.widths[1], net.widths[2] = 16, 100, 12
    assert L.mpc_mlp_workspace_bytes(ctypes.byref(net)) == 4 * (112 * 20 + 112 + 16 * 116 + 16) + 256
    assert L.mpc_mlp_linearize(ctypes.byref(net), 12, 4, 0, None, None, None, None, None, 0, None) == 0    # N = 0
    assert L.mpc_mlp_linearize(ctypes.byref(net), 12, 4, 8, None, None, None, None, None, 0, None) == -2   # x NULL
    assert L.mpc_mlp_linearize(ctypes.byref(net), 12, 4, 8, 16, 16, 16, 16, None, 0, None) == -5           # no workspace
    assert b"workspace" in L.mpc_lqr_last_error()
    assert L.mpc_mlp_linearize(ctypes.byref(net), 11, 4, 8, 16, 16, 16, 16, 16, 1 << 20, None) == -1       # widths vs n_state
    net.ctrl_carry = 4
    assert L.mpc_mlp_linearize(ctypes.byref(net), 12, 4, 8, 16, 16, 16, 16, 16, 1 << 20, None) == -5       # rollout-only option
    p = _native.Problem()
    p.B, p.T, p.ns, p.nc, p.dtype = 4, 5, 12, 4, 1
    out = _native.Outputs()
    assert L.mpc_mlp_rollout(ctypes.byref(p), None, ctypes.byref(net), None, None, None, ctypes.byref(out), None, 0, None) == -3   # fp64
    p.dtype = 0
    assert L.mpc_mlp_rollout(ctypes.byref(p), None, ctypes.byref(net), None, None, None, ctypes.byref(out), None, 0, None) == -2   # x_init NULL
    # MPC_OPT_SWEEP_ONLY needs somewhere to put the gains
    p.x_init = p.C = p.c = p.F = p.cur_x = p.cur_u = 16
    o = _native.Options()
    o.max_linesearch_iter, o.delta_u, o.flags = 10, float('nan'), _native.OPT_SWEEP_ONLY
    assert L.mpc_lqr_step(ctypes.byref(p), ctypes.byref(o), ctypes.byref(out), None, 0, 0, None) == -2
    assert b"SWEEP_ONLY" in L.mpc_lqr_last_error()


def test_simulator_entry_points_validate_arguments_without_gpu():
    L = _native.load()
    e = _native.EnvDynamics()
    e.kind, e.dt, e.u_max = 9, 0.05, 2.0
    assert L.mpc_env_linearize(ctypes.byref(e), 0, 10, None, None, None, None, None) == -5      # unknown kind
    e.kind = _native.ENV_PENDULUM
    assert L.mpc_env_linearize(ctypes.byref(e), 0, 10, None, None, None, None, None) == -2      # params NULL
    e.params = 16
    assert L.mpc_env_linearize(ctypes.byref(e), 0, 0, None, None, None, None, None) == 0        # nothing to do
    assert L.mpc_env_linearize(ctypes.byref(e), 0, 10, None, None, None, None, None) == -2      # x NULL
    assert L.mpc_env_linearize(ctypes.byref(e), 5, 10, None, None, None, None, None) == -3      # dtype
    assert L.mpc_env_linearize(None, 0, 10, None, None, None, None, None) == -2
    p = _native.Problem()
    p.B, p.T, p.ns, p.nc, p.dtype = 4, 5, 5, 1, 0          # cart-pole sized problem, pendulum descriptor
    assert L.mpc_env_traj_cost(ctypes.byref(p), ctypes.byref(e), None, None, None) == -1
    assert b"simulator" in L.mpc_lqr_last_error()
    p.ns = 3
    assert L.mpc_env_traj_cost(ctypes.byref(p), ctypes.byref(e), None, None, None) == -2        # x_init NULL
    p.B = 0
    assert L.mpc_env_traj_cost(ctypes.byref(p), ctypes.byref(e), None, None, None) == 0
    # a simulator as true_dynamics: sizes are checked against the problem, fused kernels refuse it
    o, out = _native.Options(), _native.Outputs()
    o.max_linesearch_iter, o.delta_u = 10, float('nan')
    o.true_dynamics = ctypes.pointer(e)
    p.B, p.ns, p.nc, p.T = 2, 12, 4, 3
    p.x_init = p.C = p.c = p.F = p.cur_x = p.cur_u = 16
    assert L.mpc_lqr_step(ctypes.byref(p), ctypes.byref(o), ctypes.byref(out), None, 0, 0, None) == -1
    assert L.mpc_lqr_impl_supported(ctypes.byref(p), None, 4) == 0 and L.mpc_lqr_impl_supported(ctypes.byref(p), None, 3) == 1
    p.ns, p.nc = 3, 1
    assert L.mpc_lqr_impl_supported(ctypes.byref(p), None, 4) == 1


def test_product_refuses_cpu_tensors():
    """No CPU / eager fallback: the shipped backend raises on host tensors."""
    be = _native.HipBackend()
    z = golden("step_unbounded_f64")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        be.lqr_step(tt(z, "x_init"), tt(z, "C"), tt(z, "c"), tt(z, "F"), tt(z, "f"), tt(z, "cur_x"),
                    tt(z, "cur_u"), _native.StepOptions())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mpc.MPC(3, 4, 5)(tt(z, "x_init"), QuadCost(tt(z, "C"), tt(z, "c")), LinDx(tt(z, "F"), tt(z, "f")))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setenv("MPC_LQR_HIP_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_native, "_lib", None)
    with pytest.raises(_native.NativeLibraryMissing, match="no CPU fallback"):
        _native.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mpc.pytorch_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in src.replace("oracle-backed", "").replace("an oracle", "") or fn == "_native.py", fn
    src = open(os.path.join(pkg, "mpc", "_native.py")).read()
    assert "import oracle" not in src and "from oracle" not in src and "lqr_oracle" not in src


# ---------------------------------------------------------------------------------------------
# MPC.forward host logic against the reference's own solves (kernels replaced by the oracle)
# ---------------------------------------------------------------------------------------------
MPC_CASES = ["mpc_notebook_tvlq", "mpc_linear_unbounded_big_bounds", "mpc_linear_unbounded_none",
             "mpc_linear_bounded", "mpc_linear_bounded_delta", "mpc_singleton_big_bounds", "mpc_singleton_none"]


def run_mpc_golden(z, verbose=-1, device=None):
    ns, nc, T, B = (int(v) for v in z["meta"])
    kw = {k[3:]: z[k][0] for k in z if k.startswith("kw_")}
    kw2 = dict(lqr_iter=int(kw["lqr_iter"]), exit_unconverged=bool(kw["exit_unconverged"]))
    if "delta_u" in kw:
        kw2["delta_u"] = float(kw["delta_u"])
    mv = (lambda t: t if t is None or device is None else t.to(device))
    ctrl = mpc.MPC(ns, nc, T, u_lower=mv(tt(z, "u_lower")), u_upper=mv(tt(z, "u_upper")), verbose=verbose,
                   backprop=False, **kw2)
    return ctrl(mv(tt(z, "x_init")), QuadCost(mv(tt(z, "C")), mv(tt(z, "c"))), LinDx(mv(tt(z, "F")), mv(tt(z, "f"))))


def run_ilqr_golden(z, kind, device=None, shipped=False):
    """BASELINE.json configs 2 / 3 at small batch: iLQR on simulator dynamics, AUTO_DIFF.
    shipped=False: plain torch modules (tests/envs.py) -> autograd linearisation + host-driven module
    rollout; shipped=True: mpc.env_dx modules -> linearisation kernel + simulator inside the rollout."""
    import envs
    ns, nc, T, B, lqr_iter = (int(v) for v in z["meta"])
    if shipped:
        from mpc.env_dx import cartpole, pendulum
        dx = pendulum.PendulumDx() if kind == "pendulum" else cartpole.CartpoleDx()
        dx.params = dx.params.double()
    else:
        dx = (envs.PendulumSim if kind == "pendulum" else envs.CartpoleSim)()
    mv = (lambda t: t if device is None else t.to(device))
    ctrl = mpc.MPC(ns, nc, T, u_lower=float(z["lower"][0]), u_upper=float(z["upper"][0]), lqr_iter=lqr_iter,
                   verbose=-1, exit_unconverged=False, detach_unconverged=False,
                   linesearch_decay=float(z["decay"][0]), max_linesearch_iter=int(z["max_ls"][0]),
                   grad_method=mpc.GradMethods.AUTO_DIFF, eps=float(z["eps"][0]))
    return ctrl(mv(tt(z, "x_init")), QuadCost(mv(tt(z, "Q")), mv(tt(z, "p"))), dx)


@pytest.mark.parametrize("kind", ["pendulum", "cartpole"])
def test_ilqr_on_simulator_dynamics_matches_reference(kind, oracle_backend):
    """The reference's own PendulumDx / CartpoleDx solves (mpc/env_dx/*.py, 8 iLQR iterations, B = 4,
    float64): same trajectories and costs from this package's driver + fresh dynamics modules."""
    z = golden("ilqr_%s_f64" % kind)
    x, u, costs = run_ilqr_golden(z, kind)
    np.testing.assert_allclose(costs.detach().numpy(), z["costs"], rtol=1e-5)
    np.testing.assert_allclose(x.detach().numpy(), z["x"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(u.detach().numpy(), z["u"], rtol=1e-4, atol=1e-4)
    assert "lqr_sweep" in oracle_backend.calls          # module dynamics: sweep on the kernel, rollout on the host


@pytest.mark.parametrize("kind", ["pendulum", "cartpole"])
def test_ilqr_on_shipped_simulators_takes_the_kernel_path(kind, oracle_backend):
    """mpc.env_dx.PendulumDx / CartpoleDx: the driver linearises them with the closed-form kernel
    and rolls them out inside the step kernel (here: their oracle stand-ins) -- same solves."""
    z = golden("ilqr_%s_f64" % kind)
    x, u, costs = run_ilqr_golden(z, kind, shipped=True)
    np.testing.assert_allclose(costs.detach().numpy(), z["costs"], rtol=1e-5)
    np.testing.assert_allclose(x.detach().numpy(), z["x"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(u.detach().numpy(), z["u"], rtol=1e-4, atol=1e-4)
    assert "inline_linearize" in oracle_backend.calls and "lqr_sweep" not in oracle_backend.calls


def test_shipped_simulator_modules_mirror_the_reference():
    """forward of mpc.env_dx modules == the reference modules' outputs (tests/golden/env_*.npz);
    attributes the examples read are there."""
    from mpc.env_dx import cartpole, pendulum
    for name, dx in (("env_pendulum_f64", pendulum.PendulumDx()),
                     ("env_pendulum_full_f64", pendulum.PendulumDx(params=torch.tensor([9.0, 1.2, 0.8, 0.3, 0.2]), simple=False)),
                     ("env_cartpole_f64", cartpole.CartpoleDx())):
        z = golden(name)
        ns, nc, T, B = (int(v) for v in z["meta"])
        dx.params = torch.from_numpy(z["params"])
        nxt = dx(torch.from_numpy(z["x"][:-1]).reshape(-1, ns), torch.from_numpy(z["u"][:-1]).reshape(-1, nc))
        np.testing.assert_allclose(nxt.numpy(), z["next"].reshape(-1, ns), rtol=1e-12, atol=1e-12)
        assert dx(torch.from_numpy(z["x"][0, 0]), torch.from_numpy(z["u"][0, 0])).shape == (ns,)
        q, p = dx.get_true_obj()
        assert q.shape == (ns + nc,) and p.shape == (ns + nc,)
        for attr in ("lower", "upper", "mpc_eps", "linesearch_decay", "max_linesearch_iter", "goal_state",
                     "goal_weights", "ctrl_penalty", "dt", "n_state", "n_ctrl"):
            assert hasattr(dx, attr)
        assert dx.lower == float(z["lower"][0]) and dx.linesearch_decay == float(z["decay"][0])


def test_dynamics_modules():
    """mpc.dynamics: NNDynamics.grad_input == autograd Jacobian (reference tests/test_dynamics.py:25-56),
    AffineDynamics, CtrlPassthroughDynamics."""
    from mpc.dynamics import AffineDynamics, CtrlPassthroughDynamics, NNDynamics
    torch.manual_seed(0)
    for act in ("relu", "sigmoid"):
        for passthrough in (True, False):
            net = NNDynamics(4, 2, hidden_sizes=[16, 8], activation=act, passthrough=passthrough).double()
            x = torch.randn(5, 4, dtype=torch.float64, requires_grad=True)
            u = torch.randn(5, 2, dtype=torch.float64, requires_grad=True)
            y = net(x, u)
            R, S = net.grad_input(x, u)
            for j in range(4):
                gx, gu = torch.autograd.grad(y[:, j].sum(), [x, u], retain_graph=True)
                np.testing.assert_allclose(R[:, j].detach().numpy(), gx.numpy(), atol=1e-12)
                np.testing.assert_allclose(S[:, j].detach().numpy(), gu.numpy(), atol=1e-12)
            assert net(x[0], u[0]).shape == (4,)
    A, Bm, cc = torch.randn(3, 3), torch.randn(3, 2), torch.randn(3)
    aff = AffineDynamics(A, Bm, cc)
    x, u = torch.randn(6, 3), torch.randn(6, 2)
    np.testing.assert_allclose(aff(x, u).numpy(), (x @ A.t() + u @ Bm.t() + cc).numpy(), rtol=1e-6)
    R, S = aff.grad_input(x, u)
    assert R.shape == (6, 3, 3) and S.shape == (6, 3, 2)
    pas = CtrlPassthroughDynamics(aff)
    tx = torch.cat((torch.randn(6, 2), x), 1)
    out = pas(tx, u)
    np.testing.assert_allclose(out[:, :2].numpy(), u.numpy())
    np.testing.assert_allclose(out[:, 2:].numpy(), aff(x, u).numpy())


def run_slew_golden(z, device=None):
    from mpc.dynamics import NNDynamics
    ns, nc, T, B = (int(v) for v in z["meta"])
    mv = (lambda t: t if device is None else t.to(device))
    dyn = NNDynamics(ns, nc, [10, 10], activation="sigmoid").double()
    with torch.no_grad():
        for i, fc in enumerate(dyn.fcs):
            fc.weight.copy_(torch.from_numpy(z["W%d" % i]))
            fc.bias.copy_(torch.from_numpy(z["b%d" % i]))
    if device is not None:
        dyn = dyn.to(device)
        dyn._wire()
    C, c, x0 = (mv(tt(z, k)).requires_grad_(True) for k in ("C", "c", "x_init"))
    prev = mv(tt(z, "prev_ctrl")) if "prev_ctrl" in z else None
    ctrl = mpc.MPC(ns, nc, T, mv(tt(z, "lo")), mv(tt(z, "hi")), None, lqr_iter=40, verbose=-1,
                   max_linesearch_iter=1, grad_method=GradMethods.ANALYTIC,
                   slew_rate_penalty=float(z["gamma"][0]), prev_ctrl=prev, exit_unconverged=False)
    x, u, costs = ctrl(x0, QuadCost(C, c), dyn)
    loss = (x * mv(tt(z, "wx"))).sum() + (u * mv(tt(z, "wu"))).sum()
    gC, gc, gx0, gb0 = torch.autograd.grad(loss, [C, c, x0, dyn.fcs[0].bias])
    return x, u, costs, gC, gc, gx0, gb0


MODULE_COST_CASES = ["mpc_module_cost_f64", "mpc_module_cost_wide_f64"]


def run_module_cost_golden(z, device=None):
    """MPC.forward with a non-quadratic nn.Module cost (tests/envs.py SmoothCost) on LinDx dynamics."""
    import envs
    ns, nc, T, B = (int(v) for v in z["meta"])
    mv = (lambda t: t if device is None else t.to(device))
    cost = envs.SmoothCost(ns + nc, seed=int(z["seed"][0]))
    if device is not None:
        cost = cost.to(device)
    bound = float(z["bound"][0])
    ctrl = mpc.MPC(ns, nc, T, u_lower=-bound, u_upper=bound, lqr_iter=int(z["lqr_iter"][0]), verbose=-1, n_batch=B,
                   exit_unconverged=False, detach_unconverged=False, eps=1e-9)
    x, u, costs = ctrl(mv(tt(z, "x_init")), cost, LinDx(mv(tt(z, "F")), mv(tt(z, "f"))))
    loss = (x * mv(tt(z, "wx"))).sum() + (u * mv(tt(z, "wu"))).sum()
    g_goal, g_P = torch.autograd.grad(loss, [cost.goal, cost.P])
    Cq, cq, sc = ctrl.approximate_cost(mv(tt(z, "x")), mv(tt(z, "u")), cost, diff=False)
    return x, u, costs, g_goal, g_P, Cq, cq, sc


def check_module_cost(out, z, tol):
    x, u, costs, g_goal, g_P, Cq, cq, sc = (t.detach().cpu().numpy() for t in out)
    # the expansion itself (mpc/mpc.py:447-487) at the reference's solution: autograd of the same module
    np.testing.assert_allclose(Cq, z["approx_C"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(cq, z["approx_c"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(sc, z["approx_costs"], rtol=1e-12)
    np.testing.assert_allclose(u, z["u"], rtol=tol, atol=tol)
    np.testing.assert_allclose(x, z["x"], rtol=tol, atol=tol)
    np.testing.assert_allclose(costs, z["costs"], rtol=max(tol, 1e-7))
    for g, k in ((g_goal, "g_goal"), (g_P, "g_P")):
        np.testing.assert_allclose(g, z[k], rtol=10 * tol, atol=10 * tol * (1 + np.abs(z[k]).max()))


@pytest.mark.parametrize("name", MODULE_COST_CASES)
@pytest.mark.parametrize("lockstep", [True, False])
def test_module_cost_matches_reference(name, lockstep):
    """SURVEY.md 8(f-4), `approximate_cost`: the reference's MPC.forward on a non-quadratic module cost (goldens
    made by tests/golden/make_golden.py module_cost_case) -- same expansion, same solve, same gradients into the
    module's parameters.  lockstep = the reference's batch-global pnqp / line-search loops (agreement to
    rounding); per problem = what the kernels do (within the pnqp stopping tolerance)."""
    prev = _native.set_backend_for_testing(OracleBackend(lockstep=lockstep))
    try:
        z = golden(name)
        out = run_module_cost_golden(z)
    finally:
        _native.set_backend_for_testing(prev)
    check_module_cost(out, z, 1e-7 if lockstep else 2e-4)


@pytest.mark.parametrize("name", ["mpc_slew_nn_f64", "mpc_slew_nn_prev_f64"])
@pytest.mark.parametrize("lockstep", [True, False])
def test_slew_rate_penalty_matches_reference(name, lockstep):
    """slew_rate_penalty (+ prev_ctrl) on an NNDynamics module with box constraints -- the setting of the
    reference's tests/test_mpc.py:652-744: same solve, same gradients w.r.t. C, c, x_init and a weight
    of the dynamics network.  lockstep: the oracle replays the reference's batch-global pnqp loop ->
    agreement to rounding; per-problem (what the kernels do): within the pnqp stopping tolerance."""
    prev = _native.set_backend_for_testing(OracleBackend(lockstep=lockstep))
    try:
        z = golden(name)
        x, u, costs, gC, gc, gx0, gb0 = run_slew_golden(z)
    finally:
        _native.set_backend_for_testing(prev)
    tol = 1e-9 if lockstep else 2e-4
    np.testing.assert_allclose(u.detach().numpy(), z["u"], rtol=tol, atol=tol)
    np.testing.assert_allclose(x.detach().numpy(), z["x"], rtol=tol, atol=tol)
    np.testing.assert_allclose(costs.detach().numpy(), z["costs"], rtol=max(tol, 1e-6))
    for g, k in ((gC, "gC"), (gc, "gc"), (gx0, "gx0"), (gb0, "gb0")):
        np.testing.assert_allclose(g.numpy(), z[k], rtol=max(tol, 1e-7) * 10, atol=max(tol, 1e-9) * (1 + np.abs(z[k]).max()))


@pytest.mark.parametrize("name", MPC_CASES)
def test_mpc_forward_matches_reference_solves(name, oracle_backend):
    """tests/test_mpc.py:91-299 inputs + the notebook problem: same (x, u, costs) as the reference."""
    z = golden(name)
    x, u, costs = run_mpc_golden(z)
    tol = 1e-6 if z["C"].dtype == np.float64 else 2e-4
    np.testing.assert_allclose(x.detach().numpy(), z["x"], rtol=tol, atol=tol)
    np.testing.assert_allclose(u.detach().numpy(), z["u"], rtol=tol, atol=tol)
    np.testing.assert_allclose(costs.numpy(), z["costs"], rtol=1e-6)
    if "delta_u" in name:
        assert float(u.abs().max()) <= 0.1 + 1e-12          # tests/test_mpc.py:239-240
    assert "select_best" in oracle_backend.calls


def test_mpc_promises_a_symmetric_C_only_after_the_first_step_has_checked_it(oracle_backend):
    """MPC_OPT_C_SYMMETRIC (include/mpc_lqr.h): the first step of a solve runs the kernels' symmetry test on C; its verdict
    comes back with the convergence flags (bit 1 of mpc_select_best's flag word), and only then -- C does not change
    during a solve -- do the remaining steps, and the KKT backward attached at the end, carry the promise.  A step
    whose status reports MPC_ST_C_ASYMMETRIC keeps every later step on the tested path."""
    from mpc import mpc
    from mpc.mpc import LinDx, QuadCost
    g = torch.Generator().manual_seed(5)
    T, B, ns, nc = 5, 3, 3, 2
    A = torch.randn(T, B, ns + nc, ns + nc, generator=g, dtype=torch.float64)
    C = A.transpose(2, 3).matmul(A)
    c = torch.randn(T, B, ns + nc, generator=g, dtype=torch.float64)
    F = torch.cat((torch.eye(ns, dtype=torch.float64).expand(T - 1, B, ns, ns), 0.3 * torch.randn(T - 1, B, ns, nc, generator=g, dtype=torch.float64)), 3)
    x0 = torch.randn(B, ns, generator=g, dtype=torch.float64)
    ctrl = mpc.MPC(ns, nc, T, u_lower=-0.2, u_upper=0.2, lqr_iter=6, verbose=-1, exit_unconverged=False, eps=0.0)
    Cg = C.clone().requires_grad_(True)
    x, u, _ = ctrl(x0, QuadCost(Cg, c), LinDx(F, None))
    steps = [s for s in oracle_backend.calls if s.startswith("step:")]
    assert steps[:2] == ["step:c_unknown"] * 2 and len(steps) > 2 and set(steps[2:]) == {"step:c_symmetric"}
    assert ctrl._c_symmetric
    seen = {}
    real = oracle_backend.kkt_backward
    oracle_backend.kkt_backward = lambda *a, **k: (seen.setdefault("opts", a[8]), real(*a, **k))[1]
    u.sum().backward()
    assert seen["opts"].c_symmetric                     # the nested solve of the backward inherits the verdict
    # the same solve with the stand-in reporting an asymmetric C on its first step: no promise, ever
    oracle_backend.calls.clear()
    step = oracle_backend.lqr_step

    def flagged(*a, **k):
        r = step(*a, **k)
        r["status"] = r["status"] | 8
        return r
    oracle_backend.lqr_step = flagged
    ctrl(x0, QuadCost(C, c), LinDx(F, None))
    steps = [s for s in oracle_backend.calls if s.startswith("step:")]
    assert len(steps) > 2 and set(steps) == {"step:c_unknown"} and not ctrl._c_symmetric
    # (round 4) ... and a first step solved by a kernel that never TESTS C -- the generic, lane-per-problem and
    # row-per-problem kernels: a 12/4 solve with max_linesearch_iter > 16, every float64 solve -- reports no
    # MPC_ST_C_ASYMMETRIC either, which is not a verdict: without MPC_ST_C_TESTED in the status words there is no promise,
    # and the backward keeps the route that uses C as given
    oracle_backend.calls.clear()
    oracle_backend.lqr_step = step
    oracle_backend.tests_c = False
    try:
        Cg2 = C.clone().requires_grad_(True)
        x2, u2, _ = ctrl(x0, QuadCost(Cg2, c), LinDx(F, None))
        steps = [s for s in oracle_backend.calls if s.startswith("step:")]
        assert len(steps) > 2 and set(steps) == {"step:c_unknown"} and not ctrl._c_symmetric
        seen.clear()
        u2.sum().backward()
        assert not seen["opts"].c_symmetric
    finally:
        oracle_backend.tests_c = True


def run_du_norm_golden(z, reference_du_norm, device=None):
    """tests/golden/mpc_du_norm_B8_f64.npz (make_golden.py: du_norm_case): the reference's short solve of 8 problems with a loose
    eps and detach_unconverged -- forward + the gradients of the fixture's linear loss.  Returns x, u, costs, (dx_init, dC, dc)."""
    ns, nc, T, B, lqr_iter, _seed = (int(v) for v in z["meta"])
    mv = (lambda t: t if device is None else t.to(device))
    C, c, x_init = (mv(tt(z, k)).clone().requires_grad_(True) for k in ("C", "c", "x_init"))
    ctrl = mpc.MPC(ns, nc, T, u_lower=-float(z["beta"][0]), u_upper=float(z["beta"][0]), lqr_iter=lqr_iter, verbose=-1,
                   exit_unconverged=False, detach_unconverged=True, eps=float(z["eps"][0]), n_batch=B, u_init=mv(tt(z, "u_init")),
                   reference_du_norm=reference_du_norm)
    x, u, costs = ctrl(x_init, QuadCost(C, c), LinDx(mv(tt(z, "F")), mv(tt(z, "f"))))
    loss = (x * mv(tt(z, "dl_dx"))).sum() + (u * mv(tt(z, "dl_du"))).sum()
    return x, u, costs, torch.autograd.grad(loss, [x_init, C, c])


def check_du_norm_golden(z, out, atol=1e-7):
    x, u, costs, (gx0, gC, gc) = out
    h = lambda t: t.detach().cpu().numpy()
    np.testing.assert_allclose(h(x), z["x"], rtol=1e-6, atol=atol)
    np.testing.assert_allclose(h(u), z["u"], rtol=1e-6, atol=atol)
    np.testing.assert_allclose(h(costs), z["costs"], rtol=1e-7)
    # the detach mask IS the reference's: the problems whose row of the mixed-up vector is above eps get no gradient
    dead = np.array([float(np.abs(h(gc)[:, b]).max()) == 0.0 for b in range(z["keep"].shape[0])])
    assert (dead == ~z["keep"]).all(), (dead, z["keep"])
    np.testing.assert_allclose(h(gx0), z["dx_init"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(h(gC), z["dC"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(h(gc), z["dc"], rtol=1e-5, atol=1e-8)


def test_reference_du_norm_option_reproduces_the_references_detach_mask(oracle_backend):
    """VERDICT r05 missing 4: for n_batch > 1 the reference's `full_du_norm` mixes the problems of a batch (mpc/lqr_step.py:243-245,
    a transpose in front of the reshape), and that vector decides the eps exit (mpc/mpc.py:299) and which problems
    detach_unconverged cuts off (:321-334).  Default here: each problem's own norm.  `reference_du_norm=True`
    reproduces the reference: same trajectories, same mask (the fixture's is [1,0,1,0,0,0,1,1] where the per-problem norms give
    another), same gradients -- held to the unmodified reference's outputs."""
    z = golden("mpc_du_norm_B8_f64")
    assert z["keep"].any() and (~z["keep"]).any()
    check_du_norm_golden(z, run_du_norm_golden(z, True))
    assert "du_norm_reference" in oracle_backend.calls and "lqr_rollout" in oracle_backend.calls
    # the default keeps each problem's own norm: a DIFFERENT mask on this fixture (the documented deviation)
    oracle_backend.calls.clear()
    gc = run_du_norm_golden(z, False)[3][2].detach().numpy()
    dead = np.array([float(np.abs(gc[:, b]).max()) == 0.0 for b in range(z["keep"].shape[0])])
    assert (dead != ~z["keep"]).any() and "du_norm_reference" not in oracle_backend.calls
    # n_batch = 1: the reference's expression IS the problem's own norm; the option changes nothing and costs nothing
    from oracle import lqr_oracle as O
    u, nu = torch.randn(5, 1, 2, dtype=torch.float64), torch.randn(5, 1, 2, dtype=torch.float64)
    assert torch.allclose(oracle_backend.du_norm_reference(u, nu), (u - nu).pow(2).sum((0, 2)).sqrt())


def test_mpc_forward_never_writes_the_callers_u_init(oracle_backend):
    """The reference never mutates `u_init` (mpc/mpc.py:230-243 only reads it).  The ping-pong step plans of
    this package write their new controls into raw buffers: a 3-D, contiguous, right-dtype u_init must not
    be one of them -- a warm start reused over repeated solves would otherwise change between calls."""
    z = golden("mpc_linear_bounded")
    ns, nc, T, B = (int(v) for v in z["meta"])
    g = torch.Generator().manual_seed(5)
    u_init = (0.1 * torch.randn(T, B, nc, generator=g)).to(tt(z, "C").dtype).contiguous()
    keep = u_init.clone()
    outs = []
    for _ in range(2):
        ctrl = mpc.MPC(ns, nc, T, u_lower=tt(z, "u_lower"), u_upper=tt(z, "u_upper"), u_init=u_init, lqr_iter=6,
                       verbose=-1, backprop=False, exit_unconverged=False)
        outs.append(ctrl(tt(z, "x_init"), QuadCost(tt(z, "C"), tt(z, "c")), LinDx(tt(z, "F"), tt(z, "f"))))
        assert torch.equal(u_init, keep), "MPC.forward wrote into the caller's u_init"
    assert torch.equal(outs[0][1], outs[1][1])          # same warm start -> same answer


def test_known_answer_table(oracle_backend):
    """examples/Time Varying Linear-Quadratic Control.ipynb cell 1: the printed iteration table."""
    z = golden("mpc_notebook_tvlq")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        run_mpc_golden(z, verbose=1)
    txt = buf.getvalue()
    assert "Initial mean(cost): 3.9041e+01" in txt
    rows = [[c.strip() for c in line.strip("| \n").split("|")] for line in txt.splitlines()
            if re.match(r"\|\s*\d+\s*\|", line)]
    published = ["6.6806e+00", "6.4417e+00", "4.5778e+00", "4.4537e+00", "4.4527e+00"]   # notebook output
    assert [r[1] for r in rows[:5]] == published
    assert [r[3] for r in rows[:5]] == ["1.00e+00", "6.00e-01", "1.00e+00", "1.00e+00", "1.00e+00"]
    # and the table the reference printed in this container agrees on the same columns
    assert np.allclose([float(r[1]) for r in rows[:6]], z["table"][:6, 1], rtol=1e-4)


@pytest.mark.parametrize("name", ["jac_unconstrained", "jac_constrained"])
def test_autograd_jacobians_match_reference(name, oracle_backend):
    """tests/test_mpc.py:303-500: du/d{C,c,x_init,F,f} through MPC.forward + LQRStepFn.backward."""
    z = golden(name)
    ns, nc, T, B = (int(v) for v in z["meta"])
    beta = float(z["beta"][0])
    tens = [torch.from_numpy(z[k]).requires_grad_(True) for k in ("C", "c", "x_init", "F", "f")]
    C, c, x0, F, f = tens
    lo, hi = -beta * torch.ones(T, B, nc).double(), beta * torch.ones(T, B, nc).double()
    x, u, _ = mpc.MPC(ns, nc, T, lo, hi, None, lqr_iter=20, verbose=-1, exit_unconverged=False)(
        x0, QuadCost(C, c), LinDx(F, f))
    np.testing.assert_allclose(u.detach().numpy(), z["u"], atol=1e-8)
    uf = u.reshape(-1)
    for i in range(len(uf)):
        gs = torch.autograd.grad(uf[i], tens, retain_graph=True)
        for k, g in zip(("dC", "dc", "dx_init", "dF", "df"), gs):
            np.testing.assert_allclose(g.reshape(-1).numpy(), z["J_" + k][i], atol=1e-8)
    assert "kkt_backward" in oracle_backend.calls


def test_lqrstep_contract(oracle_backend):
    """Return tuple / no-op mode / empty-f convention of LQRStep (mpc/lqr_step.py:277-309, 397-402)."""
    z = golden("step_unbounded_nof_f64")
    ns, nc, T, B = (int(v) for v in z["meta"][:4])
    x0, C, c, F = (tt(z, k) for k in ("x_init", "C", "c", "F"))
    cur_x, cur_u = tt(z, "cur_x"), tt(z, "cur_u")
    step = LQRStep(ns, nc, T, true_cost=QuadCost(C, c), true_dynamics=LinDx(F, None), current_x=cur_x,
                   current_u=cur_u)
    out = step(x0, C, c, F, torch.Tensor())
    assert len(out) == 6
    new_x, new_u, n_qp, costs, fdn, mean_alpha = out
    np.testing.assert_allclose(new_u.numpy(), z["new_u_pp"], atol=1e-9)
    assert n_qp.device.type == "cpu" and n_qp.shape == (1,) and mean_alpha.ndim == 0
    # no-op forward returns the nominal and wires the backward
    Cg = C.clone().requires_grad_(True)
    noop = LQRStep(ns, nc, T, true_cost=QuadCost(Cg, c), true_dynamics=LinDx(F, None), current_x=new_x,
                   current_u=new_u, no_op_forward=True)
    x2, u2 = noop(x0, Cg, c, F, torch.Tensor())
    assert torch.equal(x2, new_x) and torch.equal(u2, new_u) and u2.requires_grad
    (gC,) = torch.autograd.grad(u2.sum(), [Cg])
    assert gC.shape == C.shape and torch.isfinite(gC).all()
    with pytest.raises(AssertionError):
        LQRStep(ns, nc, T, delta_space=False, current_x=cur_x, current_u=cur_u)(x0, C, c, F, torch.Tensor())
    with pytest.raises(AssertionError):   # delta_u without bounds: unimplemented upstream too (:195)
        LQRStep(ns, nc, T, delta_u=0.1, current_x=cur_x, current_u=cur_u)(x0, C, c, F, torch.Tensor())


def test_driver_argument_handling(oracle_backend):
    z = golden("step_cfg1_f64")
    ns, nc, T, B = (int(v) for v in z["meta"][:4])
    x0, C, c, F, f = (tt(z, k) for k in ("x_init", "C", "c", "F", "f"))
    # time/batch-invariant cost given as [n,n] / [n]: expanded views, not copies (mpc/mpc.py:207-221)
    C2, c2 = C[0, 0].contiguous(), c[0, 0].contiguous()
    ctrl = mpc.MPC(ns, nc, T, lqr_iter=3, n_batch=B, exit_unconverged=False, verbose=-1)
    x, u, costs = ctrl(x0, QuadCost(C2, c2), LinDx(F, f))
    Cx = C2.expand(T, B, ns + nc, ns + nc).contiguous()
    cx = c2.expand(T, B, ns + nc).contiguous()
    x3, u3, _ = mpc.MPC(ns, nc, T, lqr_iter=3, exit_unconverged=False, verbose=-1)(x0, QuadCost(Cx, cx), LinDx(F, f))
    np.testing.assert_allclose(u.detach().numpy(), u3.detach().numpy(), atol=1e-12)
    with pytest.raises(ValueError, match="batch size"):
        mpc.MPC(ns, nc, T)(x0, QuadCost(C2, c2), LinDx(F, f))
    with pytest.raises(AssertionError):
        mpc.MPC(ns, nc, T, u_lower=-1.0)
    with pytest.raises(AssertionError):
        mpc.MPC(ns, nc, T, max_linesearch_iter=0)
    # exit_unconverged=True (the default) -> the reference's `assert False`, here a named AssertionError
    with pytest.raises(mpc.UnconvergedError):
        mpc.MPC(ns, nc, T, u_lower=tt(z, "u_lower"), u_upper=tt(z, "u_upper"), lqr_iter=1, verbose=-1)(
            x0, QuadCost(C, c), LinDx(F, f))
    # warm start through u_init [T, nc]
    u0 = torch.zeros(T, nc, dtype=x0.dtype)
    mpc.MPC(ns, nc, T, u_init=u0, lqr_iter=2, exit_unconverged=False, verbose=-1)(x0, QuadCost(C, c), LinDx(F, f))


def test_util_mirror(oracle_backend):
    z = golden("traj_cost")
    T, B, nc = z["u"].shape
    dx = LinDx(tt(z, "F"), tt(z, "f"))
    x = util.get_traj(T, tt(z, "u"), tt(z, "x_init"), dx)
    np.testing.assert_allclose(x.numpy(), z["x"], atol=1e-12)
    cost = util.get_cost(T, tt(z, "u"), QuadCost(tt(z, "C"), tt(z, "c")), dx, x_init=tt(z, "x_init"))
    np.testing.assert_allclose(cost.numpy(), z["cost"], rtol=1e-12)
    cost2 = util.get_cost(T, tt(z, "u"), QuadCost(tt(z, "C"), tt(z, "c")), x=x)
    np.testing.assert_allclose(cost2.numpy(), z["cost"], rtol=1e-12)
    a, b = torch.randn(3, 4, 5), torch.randn(3, 5)
    assert torch.allclose(util.bmv(a, b), a.bmm(b.unsqueeze(2)).squeeze(2))
    assert torch.allclose(util.bger(b, b), b.unsqueeze(2) * b.unsqueeze(1))
    v = torch.tensor([[-2.0, 0.5, 3.0]])
    assert util.eclamp(v, -1.0, 1.0).tolist() == [[-1.0, 0.5, 1.0]] and v.tolist() == [[-1.0, 0.5, 1.0]]


def test_pnqp_mirror(oracle_backend, capsys):
    from mpc import pnqp as pnqp_mod
    z = golden("pnqp_n4_warm_f64")
    x, fac, If, n_it = pnqp_mod.pnqp(tt(z, "H"), tt(z, "q"), tt(z, "lower"), tt(z, "upper"), x_init=tt(z, "x0"))
    np.testing.assert_allclose(x.numpy(), z["x_pp"], atol=1e-10)
    assert isinstance(fac, tuple) and len(fac) == 2 and n_it == int(z["iters_pp"].max())
    assert np.array_equal(If.numpy(), z["If_pp"])
    # the iteration count: a 1-element CPU tensor (as n_total_qp_iter of LQRStep), read from the device only when looked at
    assert n_it + 1 == int(z["iters_pp"].max()) + 1 and "%d" % n_it == str(int(n_it)) and n_it.shape == (1,)
    assert len(range(int(n_it))) == int(n_it)
    # (LU, pivots) solve H_ like the reference's H_lu_ does (mpc/pnqp.py:53-54)
    rhs = torch.ones(fac[0].shape[0], fac[0].shape[1], 1, dtype=fac[0].dtype)
    sol = torch.linalg.lu_solve(fac[0], fac[1], rhs)
    Hn, Ifb = z["H"], z["If_pp"].astype(bool)
    Hfree = np.where(Ifb[:, :, None] & Ifb[:, None, :], Hn, 0.0) + 1e-11 * np.eye(Hn.shape[1])
    np.testing.assert_allclose(np.einsum("bij,bjk->bik", Hfree, sol.numpy()), rhs.numpy(), atol=1e-8)


def test_module_dynamics_rollout_equals_lindx(oracle_backend):
    """Affine dynamics given as an nn.Module take the host-driven rollout; results must equal the
    LinDx kernel path (the reference's own cross-check, tests/test_mpc.py:503-558)."""
    z = golden("mpc_linear_bounded")
    ns, nc, T, B = (int(v) for v in z["meta"])
    x0, C, c, F, f = (tt(z, k) for k in ("x_init", "C", "c", "F", "f"))

    class Affine(torch.nn.Module):
        def forward(self, x, u):
            return x @ F[0, 0, :, :ns].t() + u @ F[0, 0, :, ns:].t() + f[0, 0]

        def grad_input(self, x, u):
            n = x.shape[0]
            return F[0, 0, :, :ns].expand(n, ns, ns), F[0, 0, :, ns:].expand(n, ns, nc)

    for gm in (GradMethods.ANALYTIC, GradMethods.AUTO_DIFF, GradMethods.FINITE_DIFF):
        x, u, _ = mpc.MPC(ns, nc, T, u_lower=tt(z, "u_lower"), u_upper=tt(z, "u_upper"), lqr_iter=20,
                          grad_method=gm, exit_unconverged=False, verbose=-1, backprop=False)(
            x0, QuadCost(C, c), Affine())
        np.testing.assert_allclose(u.detach().numpy(), z["u"], atol=2e-6)


def test_jacobian_helper():
    """mpc.util.jacobian against a closed form (the finite-difference helper of the reference's util.py; `torch_numdiff`, which
    SURVEY.md 2 marks out of scope and nothing in mpc/ uses, is not mirrored)."""
    A = torch.randn(2, 3, dtype=torch.float64)
    J = util.jacobian(lambda z: A @ z + z[0] * z[1], torch.tensor([[0.3, -0.2, 0.5]], dtype=torch.float64), 1e-5)
    want = A.clone(); want[:, 0] += -0.2; want[:, 1] += 0.3
    np.testing.assert_allclose(J.numpy(), want.numpy(), atol=1e-8)
    x = torch.randn(4, 3, dtype=torch.float64, requires_grad=True)
    assert util.data_maybe(None) is None and not util.data_maybe(x).requires_grad


def _slew_rate_properties(device=None):
    """reference tests/test_mpc.py:802-861 (test_lqr_slew_rate), restated."""
    from mpc.dynamics import AffineDynamics
    mv = (lambda t: t if device is None else t.to(device))
    n_batch, n_state, n_ctrl, T, alpha = 2, 3, 4, 5, 0.2
    n_sc = n_state + n_ctrl
    torch.manual_seed(1)
    C = torch.randn(T, n_batch, n_sc, n_sc, dtype=torch.float64)
    C = mv(C.transpose(2, 3).matmul(C))
    c = mv(torch.randn(T, n_batch, n_sc, dtype=torch.float64))
    x_init = mv(torch.randn(n_batch, n_state, dtype=torch.float64))
    R = mv(torch.eye(n_state, dtype=torch.float64) + alpha * torch.randn(n_state, n_state, dtype=torch.float64))
    S = mv(torch.randn(n_state, n_ctrl, dtype=torch.float64))
    f = mv(torch.randn(n_state, dtype=torch.float64))
    dynamics = AffineDynamics(R, S, f)

    def solve(**kw):
        return mpc.MPC(n_state, n_ctrl, T, u_lower=None, u_upper=None, u_init=None, lqr_iter=10, backprop=False,
                       verbose=-1, exit_unconverged=False, eps=1e-4, **kw)(x_init, QuadCost(C, c), dynamics)
    x, u, objs = solve()
    x_e, u_e, _ = solve(slew_rate_penalty=1e-6)
    np.testing.assert_allclose(x.detach().cpu().numpy(), x_e.detach().cpu().numpy(), atol=1e-3)
    np.testing.assert_allclose(u.detach().cpu().numpy(), u_e.detach().cpu().numpy(), atol=1e-3)
    x_s, u_s, objs_s = solve(slew_rate_penalty=1.)
    assert bool((objs < objs_s).all())
    assert torch.norm(u_s[:-1] - u_s[1:]).item() < torch.norm(u[:-1] - u[1:]).item()


def test_slew_rate_properties(oracle_backend):
    _slew_rate_properties()


def test_gradient_through_affine_module_equals_lindx(oracle_backend):
    """reference tests/test_mpc.py:503-558: du/dF is the same whether the dynamics come as LinDx(F) or as
    an AffineDynamics module built on the same tensor (ANALYTIC linearisation)."""
    from mpc.dynamics import AffineDynamics
    npr = np.random.RandomState(0)
    torch.manual_seed(0)
    n_batch, n_state, n_ctrl, T = 1, 2, 2, 2
    n_sc = n_state + n_ctrl
    C = 10. * npr.randn(T, n_batch, n_sc, n_sc)
    C = torch.tensor(np.matmul(C.transpose(0, 1, 3, 2), C), requires_grad=True)
    c = torch.tensor(10. * npr.randn(T, n_batch, n_sc), requires_grad=True)
    x_init = torch.tensor(npr.randn(n_batch, n_state), requires_grad=True)
    beta = 2.0
    lo, hi = -beta * torch.ones(T, n_batch, n_ctrl, dtype=torch.float64), beta * torch.ones(T, n_batch, n_ctrl, dtype=torch.float64)
    F = torch.randn(1, 1, n_state, n_sc, dtype=torch.float64).repeat(T - 1, 1, 1, 1).requires_grad_(True)
    jac = []
    for dyn in (LinDx(F), AffineDynamics(F[0, 0, :, :n_state], F[0, 0, :, n_state:])):
        x, u, _ = mpc.MPC(n_state, n_ctrl, T, lo, hi, None, lqr_iter=20, verbose=-1)(x_init, QuadCost(C, c), dyn)
        flat = u.reshape(-1)
        jac.append(torch.stack([torch.autograd.grad(flat[i], [F], retain_graph=True)[0].reshape(-1) for i in range(len(flat))]))
    np.testing.assert_allclose(jac[0].numpy(), jac[1].numpy(), atol=1e-4)
    assert jac[0].abs().max() > 0


@pytest.mark.parametrize("hidden", [[], [7], [6, 5]])
def test_augmented_network_is_the_ctrl_passthrough_map(hidden):
    """MlpSpec.augmented() (what CtrlPassthroughDynamics.native_net hands to the kernels for the slew-rate augmentation,
    mpc/dynamics.py:131-150): evaluating the augmented weights the way the kernels do -- layers, passthrough on the state
    rows, the control written into the first `ctrl_carry` rows -- is CtrlPassthroughDynamics(NNDynamics).forward."""
    from mpc._native import MlpSpec
    from mpc.dynamics import CtrlPassthroughDynamics, NNDynamics
    torch.manual_seed(4)
    ns, nc, N = 3, 2, 9
    for passthrough in (True, False):
        net = NNDynamics(ns, nc, hidden, activation="sigmoid", passthrough=passthrough).double()
        aug = MlpSpec([l.weight.detach() for l in net.fcs], [l.bias.detach() for l in net.fcs], "sigmoid", passthrough).augmented()
        assert aug.ctrl_carry == nc and aug.n_state == ns + nc and aug.n_ctrl == nc
        z, u = torch.randn(N, nc + ns, dtype=torch.float64), torch.randn(N, nc, dtype=torch.float64)
        h = torch.cat((z, u), 1)
        for i, (W, b) in enumerate(zip(aug.weights, aug.biases)):
            h = h @ W.t() + b
            if i + 1 < len(aug.weights):
                h = torch.sigmoid(h)
        out = h + (z if passthrough else 0.0)
        out[:, :nc] = u                                   # ctrl_carry rows: the control itself, no passthrough
        with torch.no_grad():
            want = CtrlPassthroughDynamics(net)(z, u)
        np.testing.assert_allclose(out.numpy(), want.numpy(), rtol=1e-12, atol=1e-12)
    # on CPU tensors (or fp64) the modules keep the host-driven path
    assert net.native_net(torch.zeros(1)) is None and CtrlPassthroughDynamics(net).native_net(torch.zeros(1)) is None


# ---------------------------------------------------------------------------------------------
# round-2 advisor findings
# ---------------------------------------------------------------------------------------------
def test_async_host_scalar_waits_wherever_it_is_buried_and_leaves_its_ring_slot():
    """n_total_qp_iter arrives by an asynchronous copy into a pinned ring slot.  Every way of handing it to torch must wait
    for the copy first -- also nested in a list (torch.stack / torch.cat) or passed by keyword -- and the value must move
    out of the slot, which a later solve reuses."""
    from mpc.lqr_step import _AsyncHostScalar

    class LateCopy:                       # stands in for the CUDA event: the "copy" lands when somebody waits for it
        def __init__(self, slot, value):
            self.slot, self.value, self.waits = slot, value, 0

        def synchronize(self):
            self.waits += 1
            self.slot.fill_(self.value)

    def fresh(value):
        slot = torch.zeros(1)
        ev = LateCopy(slot, value)
        return _AsyncHostScalar(slot, ev), slot, ev

    n, slot, ev = fresh(7.0)
    assert float(torch.stack([n])[0]) == 7.0 and ev.waits == 1
    n, slot, ev = fresh(5.0)
    assert float(torch.cat((n, torch.ones(1)))[0]) == 5.0
    n, slot, ev = fresh(3.0)
    assert float(torch.add(torch.ones(1), other=n)) == 4.0
    n, slot, ev = fresh(9.0)
    assert float(n) == 9.0 and ev.waits == 1
    slot.fill_(-1.0)                      # the ring comes round: a later solve lands in the same slot
    assert float(n) == 9.0 and n.item() == 9.0 and ev.waits == 1


def test_pnqp_iteration_count_is_a_host_scalar_that_warns_when_read(capsys):
    """mpc.pnqp's 4th return value (the reference's `i`, mpc/pnqp.py:59, 82): a 1-element CPU tensor like n_total_qp_iter;
    the "Did not converge" warning (:81) rides on the same host read -- at once for CPU tensors, at the first look otherwise."""
    from mpc.pnqp import _iteration_count
    from mpc.lqr_step import _AsyncHostScalar
    n = _iteration_count(torch.tensor([3, 7, 5]), torch.tensor([0, 0, 0]))
    assert int(n) == 7 and n / 2 == 3.5 and float(n + 1) == 8.0 and list(range(int(n)))[-1] == 6
    # ... and it goes wherever the reference's Python int goes: range(), indexing, a comparison used as a bool
    assert list(range(n))[-1] == 6 and "abcdefgh"[n] == "h" and (n == 7) and not (n == 6) and [0] * n == [0] * 7
    assert "Did not converge" not in capsys.readouterr().out
    _iteration_count(torch.tensor([19]), torch.tensor([1]))
    assert "pnqp warning: Did not converge" in capsys.readouterr().out

    class Ev:
        def synchronize(self):
            pass
    late = _AsyncHostScalar(torch.full((1,), 4.0), Ev())
    seen = []
    late._on_settle = lambda: seen.append(1)
    assert seen == [] and float(late) == 4.0 and seen == [1] and float(late * 2) == 8.0 and seen == [1]


def test_network_kernels_are_offered_only_what_they_take(monkeypatch):
    """NNDynamics.native_net: the library's own LDS-budget test decides (mpc_mlp_supported), and a module whose forward is
    not the stock one -- a subclass overriding it, a registered hook, a swapped activation -- keeps the module path."""
    from mpc import _native
    from mpc.dynamics import NNDynamics, CtrlPassthroughDynamics
    W = _native.MlpSpec.widths_supported
    assert W([16, 100, 12]) and W([16, 300, 300, 12]) and W([5, 800, 4])
    assert not W([5, 1024, 4]) and not W([20, 512, 12]) and not W([5, 2048, 4])       # the advisor's examples: too wide for the LDS
    monkeypatch.setattr(_native.MlpSpec, "supported", staticmethod(lambda weights, activation, like: True))
    like = torch.zeros(1)
    assert NNDynamics(3, 1, [8]).native_net(like) is not None

    class Normalised(NNDynamics):
        def forward(self, x, u):
            return super().forward(x / 2.0, u)
    assert Normalised(3, 1, [8]).native_net(like) is None
    hooked = NNDynamics(3, 1, [8])
    hooked.register_forward_hook(lambda m, i, o: o)
    assert hooked.native_net(like) is None
    swapped = NNDynamics(3, 1, [8])
    swapped.acts[0] = torch.tanh
    assert swapped.native_net(like) is None
    plain = NNDynamics(3, 1, [8])
    plain(torch.zeros(2, 3), torch.zeros(2, 1))
    assert len(plain.zs) == 1
    plain.native_net(like)
    assert plain.zs == []                 # the kernels do not refresh the activations grad_input re-uses
    assert CtrlPassthroughDynamics(NNDynamics(3, 1, [8])).native_net(like) is not None


def test_cached_parameter_copy_follows_edits_the_version_counter_does_not_see():
    """_native._device_copy_of keeps the device copy of a simulator's parameter block between the three requests of a solve.
    In-place edits through `.data` (or a numpy alias) leave `_version` where it was: the cache must notice them
    by content, an untouched block must keep hitting it, and the invalidate hook must empty it."""
    from mpc import _native
    t = torch.tensor([10.0, 1.0, 1.0], dtype=torch.float32)
    dev, dt = torch.device("cpu"), torch.float64           # (another dtype: the same path a host -> device copy takes)
    a = _native._device_copy_of(t, dev, dt)
    assert _native._device_copy_of(t, dev, dt) is a                                  # unchanged: the cached copy
    v = t._version
    t.data.mul_(2.0)
    assert t._version == v                                                            # the hole: no version bump ...
    b = _native._device_copy_of(t, dev, dt)
    assert b is not a and torch.equal(b, t.to(dt))                                    # ... and still the new numbers
    t.numpy()[1] = 7.0                                                                # a numpy alias
    assert float(_native._device_copy_of(t, dev, dt)[1]) == 7.0
    t.mul_(0.5)                                                                       # an ordinary in-place op
    c = _native._device_copy_of(t, dev, dt)
    assert torch.equal(c, t.to(dt)) and _native._device_copy_of(t, dev, dt) is c
    _native.invalidate_param_copies()
    assert _native._device_copy_of(t, dev, dt) is not c
    big = torch.zeros(65)
    assert _native._device_copy_of(big, dev, dt) is not _native._device_copy_of(big, dev, dt)      # never cached


def test_network_iterations_are_the_general_loop(oracle_backend, monkeypatch):
    """MPC._iterate_network (round 5: an NNDynamics the kernels take + QuadCost + ANALYTIC: linearise, sweep and the rollout
    through the network as pre-bound calls, the nominal states taken from the previous rollout) against MPC._iterate_general
    (the module called timestep by timestep, linearised through grad_input: the reference's only path, mpc/mpc.py:245-306,
    495-512, mpc/lqr_step.py:223-225) on the oracle stand-in: the same iterates, the same best trajectory."""
    from mpc import mpc as M, _native
    from mpc.dynamics import NNDynamics
    torch.manual_seed(3)
    ns, nc, T, B = 4, 2, 6, 5
    dyn = NNDynamics(ns, nc, [12], activation="sigmoid").double()
    A = torch.randn(T, B, ns + nc, ns + nc, dtype=torch.float64)
    C = A.transpose(2, 3).matmul(A) + 0.1 * torch.eye(ns + nc, dtype=torch.float64)
    c = torch.randn(T, B, ns + nc, dtype=torch.float64)
    x0 = torch.randn(B, ns, dtype=torch.float64)
    u0 = 0.2 * torch.randn(T, B, nc, dtype=torch.float64)

    def solve():
        ctrl = M.MPC(ns, nc, T, u_lower=-0.5, u_upper=0.5, lqr_iter=4, verbose=-1, exit_unconverged=False, detach_unconverged=False,
                     grad_method=M.GradMethods.ANALYTIC, backprop=False, u_init=u0.clone())
        with torch.no_grad():
            return ctrl(x0, M.QuadCost(C, c), dyn)
    oracle_backend.calls.clear()
    xg, ug, cg = solve()                                       # native_net is None on a CPU box: the general loop
    assert "network_iteration" not in oracle_backend.calls and "lqr_sweep" in oracle_backend.calls
    monkeypatch.setattr(_native.MlpSpec, "supported", staticmethod(lambda weights, activation, like: True))
    oracle_backend.calls.clear()
    xn, un, cn = solve()
    calls = oracle_backend.calls
    assert calls.count("mlp_traj_cost") == 1 and "lqr_sweep" not in calls               # one get_traj, then the rollouts' own states
    assert [k for k in calls if k.startswith("network_iteration")] == ["network_iteration", "network_iteration",
                                                                       "network_iteration:c_symmetric", "network_iteration:c_symmetric"]
    np.testing.assert_allclose(un.numpy(), ug.numpy(), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(xn.numpy(), xg.numpy(), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(cn.numpy(), cg.numpy(), rtol=1e-9)
    assert u0.abs().max() <= 0.2 * 6 and not torch.equal(un, u0)                          # the caller's u_init is never written into
