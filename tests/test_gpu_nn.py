"""GPU parity (-m gpu) of the NNDynamics kernels (csrc/nn_dynamics.hip: the network's layers on MFMA, 16 problems per
wavefront) against oracle/env_oracle.py, which is pinned on the reference's own NNDynamics (tests/golden/nn_*.npz,
tests/test_oracle_golden.py): util.get_traj through the network, MPC.linearize_dynamics(ANALYTIC), the line-searched
rollout of one LQR step -- on the fixtures themselves and at BASELINE-sized batches on random networks.

Tolerance: float32 kernels against the float64 oracle, rtol 1e-3 / atol 1e-4 on x, u, 1e-3 on costs."""
import numpy as np
import pytest
import torch

from conftest import golden
from test_gpu_fullsize import host, strict_step_check

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NN_CASES = ["nn_sigmoid_f64", "nn_relu2_f64", "nn_headline_f64", "nn_nopass_f64"]


@pytest.fixture(scope="module")
def be():
    from mpc import _native
    assert torch.cuda.is_available(), "these tests need the MI355X"
    _native.load()            # fail loudly if the extension is missing
    return _native.HipBackend()


def f32(a):
    return torch.from_numpy(np.ascontiguousarray(a)).float().to(DEV)


def spec_of(net):
    from mpc._native import MlpSpec
    return MlpSpec([f32(W) for W in net.Ws], [f32(b) for b in net.bs], net.activation, net.passthrough)


def random_net(ns, nc, hidden, act, passthrough, seed, scale=1.0):
    from oracle import env_oracle as E
    rng = np.random.RandomState(seed)
    sizes = [ns + nc] + list(hidden) + [ns]
    Ws = [scale * rng.uniform(-1, 1, (o, i)) / np.sqrt(i) for i, o in zip(sizes, sizes[1:])]
    bs = [rng.uniform(-1, 1, o) / np.sqrt(i) for i, o in zip(sizes, sizes[1:])]
    return E.Mlp(Ws, bs, act, passthrough)


@pytest.mark.parametrize("name", NN_CASES)
def test_network_kernels_on_the_reference_fixtures(be, name):
    """The three kernels on the inputs of the fixtures the reference generated: forward / Jacobian at the random
    points, the nominal trajectory, F and f along it, and the LQR step's line-searched rollout (gains from the float64
    C oracle's sweep) against the REFERENCE's own outputs."""
    from mpc._native import StepOptions
    from oracle import env_oracle as E
    from oracle import lqr_oracle as O
    z = golden(name)
    ns, nc, T, B = (int(v) for v in z["meta"][:4])
    net = E.Mlp.from_npz(z)
    sp = spec_of(net)
    # points: one "trajectory" of length 2 per point gives the forward map; linearize gives the Jacobian
    x1, _ = be.mlp_traj_cost(f32(z["px"]), f32(np.stack((z["pu"], z["pu"]))), sp)
    np.testing.assert_allclose(host(x1)[1], z["pnext"], rtol=1e-4, atol=2e-5)
    F, f = be.mlp_linearize(sp, f32(z["px"]), f32(z["pu"]))
    np.testing.assert_allclose(host(F)[:, :, :ns], z["pR"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(host(F)[:, :, ns:], z["pS"], rtol=1e-4, atol=2e-5)
    tau = np.concatenate((z["px"], z["pu"]), 1)
    np.testing.assert_allclose(host(f), z["pnext"] - np.einsum("nij,nj->ni", np.concatenate((z["pR"], z["pS"]), 2), tau),
                               rtol=1e-4, atol=5e-5)
    # nominal trajectory and its linearisation
    xs, _ = be.mlp_traj_cost(f32(z["x_init"]), f32(z["step_cur_u"]), sp)
    np.testing.assert_allclose(host(xs), z["step_cur_x"], rtol=1e-4, atol=5e-5)
    Fl, fl = be.mlp_linearize(sp, f32(z["step_cur_x"][:-1].reshape(-1, ns)), f32(z["step_cur_u"][:-1].reshape(-1, nc)))
    np.testing.assert_allclose(host(Fl).reshape(z["step_F"].shape), z["step_F"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(host(fl).reshape(z["step_f"].shape), z["step_f"], rtol=1e-4, atol=5e-5)
    # one LQR step: sweep = float64 oracle, rollout = the kernel
    bound = float(z["bound"][0])
    lo, hi = (None, None) if np.isnan(bound) else (-bound, bound)
    decay, max_ls = float(z["decay"][0]), int(z["max_ls"][0])
    o = O.lqr_step(z["x_init"], z["C"], z["c"], z["step_F"], z["step_f"], z["step_cur_x"], z["step_cur_u"], lo, hi,
                   linesearch_decay=decay, max_linesearch_iter=max_ls, return_gains=True)
    old = E.quad_cost(z["C"], z["c"], z["step_cur_x"], z["step_cur_u"])
    r = be.mlp_rollout(f32(z["x_init"]), f32(z["C"]), f32(z["c"]), f32(o["K"]), f32(o["k"]), f32(z["step_cur_x"]),
                       f32(z["step_cur_u"]), f32(old), StepOptions(u_lower=lo, u_upper=hi, linesearch_decay=decay,
                                                                  max_linesearch_iter=max_ls), sp)
    torch.cuda.synchronize()
    np.testing.assert_allclose(host(r["new_u"]), z["step_new_u"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(host(r["new_x"]), z["step_new_x"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(host(r["costs"]), z["step_costs"], rtol=1e-3)
    assert (host(r["status"]) == 0).all()
    # trajectory cost through the network
    _, c0 = be.mlp_traj_cost(f32(z["x_init"]), f32(z["step_cur_u"]), sp, C=f32(z["C"]), c=f32(z["c"]))
    np.testing.assert_allclose(host(c0), old, rtol=1e-4)


@pytest.mark.parametrize("ns,nc,hidden,act,passthrough,B,T,bound", [
    (12, 4, [100], "sigmoid", True, 4096, 50, 0.5),          # the headline shape with the reference's default network
    (12, 4, [100], "relu", True, 1000, 20, None),            # ragged last group (1000 = 62 * 16 + 8), unbounded
    (5, 1, [64, 48], "sigmoid", True, 517, 25, 2.0),         # two hidden layers, one control
    (16, 8, [256, 32, 20], "elu", False, 260, 12, 1.0),      # n = 24: two input tiles; three hidden layers; no passthrough
    (3, 2, [], "sigmoid", True, 100, 6, None),               # a single Linear layer
])
def test_network_rollout_and_linearisation_at_full_batches(be, ns, nc, hidden, act, passthrough, B, T, bound):
    """Random networks at BASELINE-sized batches, every problem: the nominal trajectory, F / f at all (T-1) B points and
    the line-searched rollout of a step whose sweep the float64 C oracle did (line-search ties classified as in
    tests/test_gpu_fullsize.py)."""
    from mpc._native import StepOptions
    from oracle import env_oracle as E
    from oracle import lqr_oracle as O
    net = random_net(ns, nc, hidden, act, passthrough, seed=ns * 100 + nc, scale=0.8)
    sp = spec_of(net)
    rng = np.random.RandomState(B)
    n = ns + nc
    x0 = rng.randn(B, ns)
    u0 = 0.3 * rng.randn(T, B, nc)
    if bound is not None:
        u0 = np.clip(u0, -bound, bound)
    A = rng.randn(T, B, n, n)
    C = np.einsum("tbki,tbkj->tbij", A, A) + 0.1 * np.eye(n)
    c = rng.randn(T, B, n)
    xs = E.traj(E.MLP, x0, u0, net)
    xk, ck = be.mlp_traj_cost(f32(x0), f32(u0), sp, C=f32(C), c=f32(c))
    scale = 1.0 + np.abs(xs).max()
    assert np.abs(host(xk) - xs).max() < 2e-4 * scale
    old = E.quad_cost(C, c, xs, u0)
    np.testing.assert_allclose(host(ck), old, rtol=1e-3)
    Fl, fl = E.linearize(E.MLP, xs[:-1].reshape(-1, ns), u0[:-1].reshape(-1, nc), net)
    Fk, fk = be.mlp_linearize(sp, f32(xs[:-1].reshape(-1, ns)), f32(u0[:-1].reshape(-1, nc)))
    np.testing.assert_allclose(host(Fk), Fl, rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(host(fk), fl, rtol=1e-3, atol=1e-4 * scale)
    Fl, fl = Fl.reshape(T - 1, B, ns, n), fl.reshape(T - 1, B, ns)
    lo, hi = (None, None) if bound is None else (-bound, bound)
    decay, max_ls = 0.2, 6
    o = O.lqr_step(x0, C, c, Fl, fl, xs, u0, lo, hi, linesearch_decay=decay, max_linesearch_iter=max_ls, lockstep=False,
                   nthreads=O.max_threads(), return_gains=True)
    nx, nu, costs, full, alphas, trials, old2 = E.rollout_batched(E.MLP, net, x0, C, c, o["K"], o["k"], xs, u0, lo, hi,
                                                                   decay, max_ls)
    o.update(new_x=nx, new_u=nu, costs=costs, alphas=alphas, old_costs=old2, full_du_norm=full)
    r = be.mlp_rollout(f32(x0), f32(C), f32(c), f32(o["K"]), f32(o["k"]), f32(xs), f32(u0), f32(old2),
                       StepOptions(u_lower=lo, u_upper=hi, linesearch_decay=decay, max_linesearch_iter=max_ls), sp)
    torch.cuda.synchronize()
    ties = strict_step_check("nn_%d_%d_%s_B%d" % (ns, nc, act, B), r, o, B, rtol=1e-3, atol=2e-4 * scale, cost_rtol=1e-3,
                             have_gains=False)
    same = ~ties
    np.testing.assert_allclose(host(r["full_du_norm"])[same], full[same], rtol=2e-3, atol=2e-4)
    if bound is not None:
        assert (host(r["new_u"]) >= lo - 1e-6).all() and (host(r["new_u"]) <= hi + 1e-6).all()


def _mpc_solve(z, dyn, dev, lqr_iter=None):
    from mpc import mpc
    ns, nc, T, B, it = (int(v) for v in z["meta"])
    bound = float(z["bound"][0])
    lo, hi = (None, None) if np.isnan(bound) else (-bound, bound)
    ctrl = mpc.MPC(ns, nc, T, u_lower=lo, u_upper=hi, lqr_iter=lqr_iter or it, verbose=-1, exit_unconverged=False,
                   detach_unconverged=False, linesearch_decay=float(z["decay"][0]),
                   max_linesearch_iter=int(z["max_ls"][0]), grad_method=mpc.GradMethods.ANALYTIC,
                   u_init=f32(z["step_cur_u"]).to(dev))
    return ctrl(f32(z["x_init"]).to(dev), mpc.QuadCost(f32(z["C"]).to(dev), f32(z["c"]).to(dev)), dyn)


def _module_of(z, dev):
    from mpc.dynamics import NNDynamics
    from oracle import env_oracle as E
    net = E.Mlp.from_npz(z)
    ns, nc = (int(v) for v in z["meta"][:2])
    hidden = [W.shape[0] for W in net.Ws[:-1]]
    dyn = NNDynamics(ns, nc, hidden, activation=net.activation, passthrough=net.passthrough)
    with torch.no_grad():
        for fc, W, b in zip(dyn.fcs, net.Ws, net.bs):
            fc.weight.copy_(torch.from_numpy(W).float())
            fc.bias.copy_(torch.from_numpy(b).float())
    return dyn.to(dev)


@pytest.mark.parametrize("name", NN_CASES)
def test_mpc_forward_with_nndynamics_matches_the_reference_solve(be, name, monkeypatch):
    """mpc.MPC(...)(x_init, QuadCost, NNDynamics) on the device, float32 -- trajectory, linearisation and rollouts in the
    kernels -- against the reference's own float64 solve of the same problem (fixture `solve_*`), and against this
    package's host-driven path (the module called timestep by timestep), which must take the same iterations."""
    from mpc import _native
    z = golden(name)
    dyn = _module_of(z, DEV)
    calls = {"iterations": 0}
    orig = _native.HipBackend.plan_network_iteration

    def counted(self, *a, **k):
        run, outs, vouch = orig(self, *a, **k)

        def run2(j, stream=None):
            calls["iterations"] += 1
            return run(j, stream)
        return run2, outs, vouch
    # (round 5: the iterations are pre-bound -- linearise, sweep, rollout as three C calls each, MPC._iterate_network)
    monkeypatch.setattr(_native.HipBackend, "plan_network_iteration", counted)
    x, u, costs = _mpc_solve(z, dyn, DEV)
    torch.cuda.synchronize()
    assert calls["iterations"] >= 2                                    # the kernels ran, not the module loop
    np.testing.assert_allclose(host(costs), z["solve_costs"], rtol=2e-3)
    np.testing.assert_allclose(host(u), z["solve_u"], rtol=5e-3, atol=5e-3)
    np.testing.assert_allclose(host(x), z["solve_x"], rtol=5e-3, atol=5e-3)
    # the host-driven path of this package on the same problem
    monkeypatch.setattr(type(dyn), "native_net", lambda self, like: None)
    x2, u2, costs2 = _mpc_solve(z, dyn, DEV)
    np.testing.assert_allclose(host(costs), host(costs2), rtol=1e-3)
    # (two float32 paths, six iLQR iterations apart: the same tolerance as against the float64 solve)
    np.testing.assert_allclose(host(u), host(u2), rtol=5e-3, atol=5e-3)


def test_lqrstep_gradients_flow_to_the_network_parameters(be):
    """The differentiable path is unchanged by the kernels: with diff=True the linearisation goes through torch and a
    loss on (x, u) reaches the network's weights (tests/test_mpc.py:652-744 checks the same for its slew variant)."""
    z = golden("nn_sigmoid_f64")
    dyn = _module_of(z, DEV)
    x, u, costs = _mpc_solve(z, dyn, DEV, lqr_iter=8)
    loss = x.pow(2).sum() + u.pow(2).sum()
    g = torch.autograd.grad(loss, [dyn.fcs[0].weight, dyn.fcs[-1].bias], allow_unused=True)
    assert all(t is not None and torch.isfinite(t).all() for t in g)
    assert float(g[0].abs().max()) > 0


class _TanhDynamics(torch.nn.Module):
    """A user module the kernels know nothing about: x' = x + 0.1 tanh(A x + B u)."""

    def __init__(self, ns, nc, sync=False):
        super().__init__()
        g = torch.Generator().manual_seed(3)
        self.A = torch.nn.Parameter(torch.randn(ns, ns, generator=g) / ns ** 0.5)
        self.B = torch.nn.Parameter(torch.randn(ns, nc, generator=g) / ns ** 0.5)
        self.sync = sync
        self.calls = 0
        self.hip_graph = True              # nothing in forward() depends on Python-side state: safe to replay

    def forward(self, x, u):
        self.calls += 1
        if self.sync:
            float(x.sum().item())          # a host read-back: cannot be captured in a graph
        return x + 0.1 * torch.tanh(x @ self.A.t() + u @ self.B.t())


@pytest.mark.parametrize("sync", [False, True])
def test_module_rollout_replays_a_graph_and_matches_the_eager_pass(be, sync, monkeypatch):
    """An arbitrary nn.Module as dynamics (mpc/lqr_step.py:223-225): the line-search passes of `_module_rollout` are
    replays of one captured HIP graph (opt-in: the module sets `hip_graph = True`) and give what the eager loop of
    device ops gives; a module that synchronises cannot be captured and silently keeps the eager loop."""
    from mpc import lqr_step, mpc
    ns, nc, T, B = 6, 2, 12, 64
    dyn = _TanhDynamics(ns, nc, sync=sync).to(DEV)
    g = torch.Generator().manual_seed(5)
    n = ns + nc
    A = torch.randn(T, B, n, n, generator=g)
    C = (A.transpose(2, 3) @ A + 0.1 * torch.eye(n)).to(DEV)
    c = torch.randn(T, B, n, generator=g).to(DEV)
    x0 = torch.randn(B, ns, generator=g).to(DEV)

    def solve():
        ctrl = mpc.MPC(ns, nc, T, u_lower=-1.0, u_upper=1.0, lqr_iter=4, verbose=-1, exit_unconverged=False,
                       detach_unconverged=False, grad_method=mpc.GradMethods.AUTO_DIFF, backprop=False)
        with torch.no_grad():
            return ctrl(x0, mpc.QuadCost(C, c), dyn)
    lqr_step._PASS_GRAPHS.clear()
    x1, u1, c1 = solve()
    entries = list(lqr_step._PASS_GRAPHS.values())
    assert len(entries) >= 1
    assert all((e[0] is None) == sync for e in entries)       # captured, or remembered as not capturable
    monkeypatch.setenv("MPC_NO_ROLLOUT_GRAPH", "1")
    x2, u2, c2 = solve()
    np.testing.assert_allclose(host(c1), host(c2), rtol=1e-5)
    np.testing.assert_allclose(host(u1), host(u2), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(host(x1), host(x2), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", ["mpc_slew_nn_f64", "mpc_slew_nn_prev_f64"])
def test_slew_rate_penalty_on_the_network_kernels(be, name, monkeypatch):
    """slew_rate_penalty around NNDynamics (the reference's tests/test_mpc.py:652-744 setting) in float32: the augmented
    dynamics CtrlPassthroughDynamics(NNDynamics) roll out inside the network kernel (`ctrl_carry`), and the solve agrees
    with the reference's float64 fixture and with this package's host-driven float32 path."""
    from mpc import _native, mpc
    from mpc.dynamics import NNDynamics
    z = golden(name)
    ns, nc, T, B = (int(v) for v in z["meta"])
    dyn = NNDynamics(ns, nc, [10, 10], activation="sigmoid")
    with torch.no_grad():
        for i, fc in enumerate(dyn.fcs):
            fc.weight.copy_(torch.from_numpy(z["W%d" % i]).float())
            fc.bias.copy_(torch.from_numpy(z["b%d" % i]).float())
    dyn = dyn.to(DEV)
    carried = []
    orig = _native.HipBackend.mlp_rollout

    def spy(self, *a, **k):
        carried.append(a[9].ctrl_carry)
        return orig(self, *a, **k)
    monkeypatch.setattr(_native.HipBackend, "mlp_rollout", spy)

    def solve():
        prev = f32(z["prev_ctrl"]) if "prev_ctrl" in z else None
        ctrl = mpc.MPC(ns, nc, T, f32(z["lo"]), f32(z["hi"]), None, lqr_iter=40, verbose=-1, max_linesearch_iter=1,
                       grad_method=mpc.GradMethods.ANALYTIC, slew_rate_penalty=float(z["gamma"][0]), prev_ctrl=prev,
                       exit_unconverged=False, backprop=False)
        with torch.no_grad():
            return ctrl(f32(z["x_init"]), mpc.QuadCost(f32(z["C"]), f32(z["c"])), dyn)
    x, u, costs = solve()
    torch.cuda.synchronize()
    assert carried and all(c == nc for c in carried)          # the augmented network ran in the kernel
    np.testing.assert_allclose(host(u), z["u"], rtol=5e-3, atol=5e-3)
    np.testing.assert_allclose(host(x), z["x"], rtol=5e-3, atol=5e-3)
    np.testing.assert_allclose(host(costs), z["costs"], rtol=2e-3)
    monkeypatch.setattr(NNDynamics, "native_net", lambda self, like: None)
    x2, u2, costs2 = solve()
    np.testing.assert_allclose(host(u), host(u2), rtol=5e-3, atol=5e-3)
    np.testing.assert_allclose(host(costs), host(costs2), rtol=1e-3)


@pytest.mark.parametrize("ns,nc,hidden", [(4, 4, [32]), (5, 2, [24]), (6, 2, [16, 16])])
def test_ctrl_carry_rollout_matches_the_augmented_map(be, ns, nc, hidden):
    """`MlpSpec.augmented()` (CtrlPassthroughDynamics around the network, mpc/dynamics.py:131-150) through both rollout
    kernels -- (4, 4, [32]) takes the register-resident one -- against z' = (u, net(x, u)) stepped in numpy."""
    from oracle import env_oracle as E
    net = random_net(ns, nc, hidden, "sigmoid", True, seed=7)
    aug = spec_of(net).augmented()
    rng = np.random.RandomState(3)
    T, B = 9, 37
    z0 = rng.randn(B, nc + ns)
    u = rng.randn(T, B, nc)
    zs = [z0]
    for t in range(T - 1):
        zs.append(np.concatenate((u[t], E.mlp_step(zs[t][:, nc:], u[t], net)), 1))
    zk, _ = be.mlp_traj_cost(f32(z0), f32(u), aug)
    np.testing.assert_allclose(host(zk), np.stack(zs), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("ns,nc,hidden", [(12, 4, [100]), (6, 3, [40, 24])])
def test_network_rollout_with_tensor_bounds_delta_u_and_pinned_controls(be, ns, nc, hidden):
    """The options of lqr_forward that bracket the control (mpc/lqr_step.py:197-213) through both rollout kernels:
    per-(t, b, i) bounds, delta_u around the nominal, u_zero_I -- against oracle/env_oracle.rollout_batched."""
    from mpc._native import StepOptions
    from oracle import env_oracle as E
    from oracle import lqr_oracle as O
    net = random_net(ns, nc, hidden, "sigmoid", True, seed=11, scale=0.8)
    sp = spec_of(net)
    rng = np.random.RandomState(5)
    T, B, n = 14, 203, ns + nc
    x0 = rng.randn(B, ns)
    lo = -0.3 - rng.rand(T, B, nc)
    hi = 0.3 + rng.rand(T, B, nc)
    u0 = np.clip(0.3 * rng.randn(T, B, nc), lo, hi)
    A = rng.randn(T, B, n, n)
    C = np.einsum("tbki,tbkj->tbij", A, A) + 0.1 * np.eye(n)
    c = rng.randn(T, B, n)
    xs = E.traj(E.MLP, x0, u0, net)
    Fl, fl = E.linearize(E.MLP, xs[:-1].reshape(-1, ns), u0[:-1].reshape(-1, nc), net)
    Fl, fl = Fl.reshape(T - 1, B, ns, n), fl.reshape(T - 1, B, ns)
    for delta_u, masked in ((None, False), (0.25, False), (None, True)):
        mask = (rng.rand(T, B, nc) < 0.2) if masked else None
        kw = dict(u_lower=lo, u_upper=hi) if not masked else {}
        okw = dict(kw)
        if delta_u is not None:
            okw["delta_u"] = delta_u
        if masked:
            okw["u_zero_I"] = mask
        o = O.lqr_step(x0, C, c, Fl, fl, xs, u0, linesearch_decay=0.2, max_linesearch_iter=5, lockstep=False,
                       nthreads=O.max_threads(), return_gains=True, **okw)
        nx, nu, costs, full, alphas, trials, old = E.rollout_batched(
            E.MLP, net, x0, C, c, o["K"], o["k"], xs, u0, None if masked else lo, None if masked else hi, 0.2, 5,
            delta_u=delta_u, u_zero_I=mask)
        o.update(new_x=nx, new_u=nu, costs=costs, alphas=alphas, old_costs=old, full_du_norm=full)
        opts = StepOptions(u_lower=None if masked else f32(lo), u_upper=None if masked else f32(hi), delta_u=delta_u,
                           u_zero_I=None if mask is None else torch.from_numpy(mask).to(DEV), linesearch_decay=0.2,
                           max_linesearch_iter=5)
        r = be.mlp_rollout(f32(x0), f32(C), f32(c), f32(o["K"]), f32(o["k"]), f32(xs), f32(u0), f32(old), opts, sp)
        torch.cuda.synchronize()
        scale = 1.0 + np.abs(xs).max()
        strict_step_check("nn_opts_%d_%s_%s" % (ns, delta_u, masked), r, o, B, rtol=1e-3, atol=2e-4 * scale, cost_rtol=1e-3,
                          have_gains=False)
        if masked:
            assert (host(r["new_u"])[mask] == 0).all()


@pytest.mark.parametrize("ns,nc,hidden,B,T,max_ls,decay,mult", [(12, 4, [100], 1000, 20, 10, 0.5, 3.0), (12, 4, [100], 333, 12, 16, 0.7, 8.0),
                                                               (8, 4, [32, 16], 200, 10, 3, 0.2, 60.0), (5, 2, [24], 260, 15, 10, 0.5, 8.0)])
def test_network_line_search_runs_every_depth_like_the_reference(be, ns, nc, hidden, B, T, max_ls, decay, mult):
    """The line search of lqr_forward (mpc/lqr_step.py:176-179, 247-252) through the network on costs that are NOT convex in the
    state, so that the problems of one wavefront stop at every depth between the full step and the last trial: per problem the
    first step size that did not get worse, else the last one.  Round 5: the trials are decided on J(tau') - J(nominal) summed as
    a difference, and a wavefront with ONE problem left rolls that problem's remaining trials out side by side (then replays the
    accepted one) -- the step sizes, trajectories and costs are the sequential passes' (oracle/env_oracle.py, every problem)."""
    from mpc._native import StepOptions
    from oracle import env_oracle as E
    from oracle import lqr_oracle as O
    net = random_net(ns, nc, hidden, "sigmoid", True, seed=7 * ns + nc, scale=0.8)
    sp = spec_of(net)
    rng = np.random.RandomState(B + max_ls)
    n = ns + nc
    x0 = rng.randn(B, ns)
    u0 = np.clip(0.3 * rng.randn(T, B, nc), -0.5, 0.5)
    A = rng.randn(T, B, n, n)
    C = np.einsum("tbki,tbkj->tbij", A, A) + 0.1 * np.eye(n)
    C[:, :, :ns, :ns] -= (rng.rand(1, B, 1, 1) * mult * n) * np.eye(ns)          # per problem: from convex to strongly non-convex in x
    c = rng.randn(T, B, n)
    xs = E.traj(E.MLP, x0, u0, net)
    Fl, fl = E.linearize(E.MLP, xs[:-1].reshape(-1, ns), u0[:-1].reshape(-1, nc), net)
    Csw = C.copy()
    Csw[:, :, :ns, :ns] += mult * n * np.eye(ns)                                 # (the sweep needs a convex model: gains from a regularised cost)
    o = O.lqr_step(x0, Csw, c, Fl.reshape(T - 1, B, ns, n), fl.reshape(T - 1, B, ns), xs, u0, -0.5, 0.5, linesearch_decay=decay,
                   max_linesearch_iter=max_ls, lockstep=False, nthreads=O.max_threads(), return_gains=True)
    nx, nu, costs, full, alphas, trials, old = E.rollout_batched(E.MLP, net, x0, C, c, o["K"], o["k"], xs, u0, -0.5, 0.5, decay, max_ls)
    depth = np.rint(np.log(alphas) / np.log(decay)).astype(int)
    assert len(set(depth.tolist())) >= min(max_ls, 4) - 1 and (depth == max_ls - 1).any() and (depth == 0).any(), np.bincount(depth)
    r = be.mlp_rollout(f32(x0), f32(C), f32(c), f32(o["K"]), f32(o["k"]), f32(xs), f32(u0), f32(old),
                       StepOptions(u_lower=-0.5, u_upper=0.5, linesearch_decay=decay, max_linesearch_iter=max_ls), sp)
    torch.cuda.synchronize()
    ga = host(r["alphas"]).astype(np.float64)
    gdepth = np.rint(np.log(ga) / np.log(decay)).astype(int)
    # a trial whose cost ties with the nominal's to float32 rounding may fall either way (tests/test_gpu_fullsize.py): counted
    tie = gdepth != depth
    margin = np.abs(trials - old[None]) / np.maximum(1.0, np.abs(old))[None]
    for b_ in np.nonzero(tie)[0]:
        lo_, hi_ = sorted((gdepth[b_], depth[b_]))
        assert margin[lo_, b_] < 1e-4, ("problem %d: step size %g for %g without a tie" % (b_, ga[b_], alphas[b_]), margin[:, b_])
    assert tie.sum() <= max(2, B // 100), tie.sum()
    same = ~tie
    np.testing.assert_allclose(ga[same], alphas[same], rtol=1e-5)
    scale = 1.0 + np.abs(nx).max()
    np.testing.assert_allclose(host(r["new_u"])[:, same], nu[:, same], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(host(r["new_x"])[:, same], nx[:, same], rtol=1e-3, atol=2e-4 * scale)
    np.testing.assert_allclose(host(r["costs"])[same], costs[same], rtol=1e-3, atol=1e-3 * np.abs(old).max() * 1e-3)
    np.testing.assert_allclose(host(r["full_du_norm"]), full, rtol=2e-3, atol=2e-4)
