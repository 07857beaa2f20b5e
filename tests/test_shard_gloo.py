"""The N > 1 path on CPU: two processes over gloo, each solving its block of the batch (through the
oracle-backed test backend -- there is no GPU here) and one all-gather reassembling the trajectories.
On the GPU box bench.py runs the same code over RCCL with the HIP backend."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, B, out_dir):
    for p in (os.path.join(ROOT, "mpc.pytorch_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from mpc import _native, shard
    from mpc._native import StepOptions
    from oracle_backend import OracleBackend
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _native.set_backend_for_testing(OracleBackend())
    g = torch.Generator().manual_seed(0)
    T, ns, nc = 6, 4, 2
    n = ns + nc
    A = torch.randn(T, B, n, n, generator=g, dtype=torch.float64)
    C = A.transpose(2, 3).matmul(A)
    c = torch.randn(T, B, n, generator=g, dtype=torch.float64)
    F = torch.cat((torch.eye(ns).double() + 0.1 * torch.randn(T - 1, B, ns, ns, generator=g, dtype=torch.float64),
                   torch.randn(T - 1, B, ns, nc, generator=g, dtype=torch.float64)), 3)
    f = 0.1 * torch.randn(T - 1, B, ns, generator=g, dtype=torch.float64)
    x_init = torch.randn(B, ns, generator=g, dtype=torch.float64)
    cur_u = torch.zeros(T, B, nc, dtype=torch.float64)
    lo = -torch.rand(T, B, nc, generator=g, dtype=torch.float64)
    hi = torch.rand(T, B, nc, generator=g, dtype=torch.float64)
    from mpc import util
    from mpc.mpc import LinDx
    cur_x = util.get_traj(T, cur_u, x_init, LinDx(F, f))
    opts = StepOptions(u_lower=lo, u_upper=hi)
    r = shard.lqr_step_sharded(x_init, C, c, F, f, cur_x, cur_u, opts)
    ref = _native.backend().lqr_step(x_init, C, c, F, f, cur_x, cur_u, opts)      # the whole batch, locally
    # the pre-sharded entry: this rank hands over ONLY its block (clones: nothing of the whole batch is reachable from them)
    a, b = shard.shard_bounds(B, rank, world)
    cut = lambda t, dim: t.narrow(dim, a, b - a).clone()
    rp = shard.lqr_step_sharded(cut(x_init, 0), cut(C, 1), cut(c, 1), cut(F, 1), cut(f, 1), cut(cur_x, 1), cut(cur_u, 1),
                                StepOptions(u_lower=cut(lo, 1), u_upper=cut(hi, 1)), presharded=True, n_batch=B)
    # the kernel's outputs ARE the rank's slot of the collective's buffer (no packing pass), and every rank's block is
    # readable from the slots without the assembling pass
    vx, vu, vs = rp["slots"].views(rank)
    own = (b == a) or (rp["local"]["new_x"].data_ptr() == vx.data_ptr() and rp["local"]["new_u"].data_ptr() == vu.data_ptr())
    blocks_ok = all(torch.equal(rp["slots"].views(q)[0], rp["new_x"][:, shard.shard_bounds(B, q, world)[0]:shard.shard_bounds(B, q, world)[1]])
                    for q in range(world))
    bad = 0
    try:        # a block that is not the one shard_bounds gives this rank is refused (every rank takes this branch: no collective inside)
        shard.lqr_step_sharded(x_init, C, c, F, f, cur_x, cur_u, opts, presharded=True, n_batch=4 * B + 2 * world)
    except ValueError:
        bad = 1
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), block=np.array(r["block"]), own=int(own), blocks_ok=int(blocks_ok), refused=bad,
             **{k: r[k].numpy() for k in ("new_x", "new_u", "costs", "full_du_norm", "alphas")},
             **{"pre_" + k: rp[k].numpy() for k in ("new_x", "new_u", "costs", "full_du_norm", "alphas")},
             **{"ref_" + k: ref[k].numpy() for k in ("new_x", "new_u", "costs", "full_du_norm", "alphas")})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 5, 1])
def test_two_ranks_shard_the_batch_and_gather_once(tmp_path, B):
    """B = 5 leaves the ranks with blocks of 3 and 2 problems: the slots of the gather are sized for the larger one.  B = 1 is
    fewer problems than ranks: rank 1 solves nothing and still takes part in the collective.  Both entries -- full-batch
    tensors cut by the rank, and (round 4) the rank's own block handed over pre-sharded -- against the whole batch solved
    locally."""
    world = 2
    port = 29500 + (os.getpid() % 2000) + B
    mp.spawn(_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    blocks = [tuple(g["block"]) for g in got]
    assert blocks[0][0] == 0 and blocks[0][1] == blocks[1][0] and blocks[1][1] == B
    for g in got:
        for k in ("new_x", "new_u", "costs", "full_du_norm", "alphas"):
            np.testing.assert_allclose(g[k], g["ref_" + k], rtol=1e-12, atol=1e-12, err_msg=k)
            np.testing.assert_allclose(g["pre_" + k], g["ref_" + k], rtol=1e-12, atol=1e-12, err_msg="pre-sharded " + k)
        assert int(g["own"]) == 1 and int(g["blocks_ok"]) == 1 and int(g["refused"]) == 1


def test_shard_bounds_cover_the_batch():
    sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd"))
    from mpc import shard
    for B in (1, 7, 4096, 4099):
        for world in (1, 2, 3, 8):
            cuts = [shard.shard_bounds(B, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1


def _mpc_worker(rank, world, port, lockstep, out_dir):
    for p in (os.path.join(ROOT, "mpc.pytorch_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from mpc import _native, mpc, shard
    from mpc.mpc import LinDx, QuadCost
    from oracle_backend import OracleBackend
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _native.set_backend_for_testing(OracleBackend())
    g = torch.Generator().manual_seed(1)
    T, ns, nc, B = 5, 3, 2, 5
    n = ns + nc
    A = torch.randn(T, B, n, n, generator=g, dtype=torch.float64)
    C = A.transpose(2, 3).matmul(A)
    c = torch.randn(T, B, n, generator=g, dtype=torch.float64)
    F = torch.cat((torch.eye(ns).double() + 0.1 * torch.randn(T - 1, B, ns, ns, generator=g, dtype=torch.float64),
                   torch.randn(T - 1, B, ns, nc, generator=g, dtype=torch.float64)), 3)
    x_init = torch.randn(B, ns, generator=g, dtype=torch.float64)
    lo = -torch.rand(T, B, nc, generator=g, dtype=torch.float64)
    hi = torch.rand(T, B, nc, generator=g, dtype=torch.float64)
    iters = []

    class Counting(OracleBackend):
        def select_best(self, *a, **k):
            iters.append(1)
            return super().select_best(*a, **k)
    _native.set_backend_for_testing(Counting())
    ctrl = mpc.MPC(ns, nc, T, u_lower=lo, u_upper=hi, lqr_iter=30, verbose=-1, exit_unconverged=False, eps=1e-6)
    x, u, costs = shard.mpc_forward_sharded(ctrl, x_init, QuadCost(C, c), LinDx(F), lockstep=lockstep)
    n_shard = len(iters)
    # the pre-sharded entry (round 4): the rank hands over its own block of everything, the controller's tensor bounds included
    a, b = shard.shard_bounds(B, rank, world)
    cut = lambda t, dim: t.narrow(dim, a, b - a).clone()
    ctrl_p = mpc.MPC(ns, nc, T, u_lower=cut(lo, 1), u_upper=cut(hi, 1), lqr_iter=30, verbose=-1, exit_unconverged=False, eps=1e-6)
    xp, up, cp = shard.mpc_forward_sharded(ctrl_p, cut(x_init, 0), QuadCost(cut(C, 1), cut(c, 1)), LinDx(cut(F, 1)), lockstep=lockstep,
                                           presharded=True, n_batch=B)
    del iters[:]
    xr, ur, cr = ctrl(x_init, QuadCost(C, c), LinDx(F))              # the whole batch on one rank
    np.savez(os.path.join(out_dir, "mpc_rank%d.npz" % rank), x=x.numpy(), u=u.numpy(), costs=costs.numpy(),
             xp=xp.numpy(), up=up.numpy(), cp=cp.numpy(),
             xr=xr.detach().numpy(), ur=ur.detach().numpy(), cr=cr.detach().numpy(), n_shard=n_shard, n_full=len(iters))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("lockstep", [False, True])
def test_sharded_mpc_forward(tmp_path, lockstep):
    """MPC.forward over two ranks (blocks of 3 and 2 problems): same trajectories as the whole batch on one
    rank; in lock-step mode the shards also run exactly the iterations of the whole-batch solve (the
    batch-wide stop test of mpc/mpc.py:299 is all-reduced), otherwise each shard stops on its own."""
    world = 2
    port = 31500 + (os.getpid() % 2000) + int(lockstep)
    mp.spawn(_mpc_worker, args=(world, port, lockstep, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(os.path.join(str(tmp_path), "mpc_rank%d.npz" % r)) for r in range(world)]
    for g in got:
        np.testing.assert_allclose(g["u"], g["ur"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(g["x"], g["xr"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(g["costs"], g["cr"], rtol=1e-6)
        np.testing.assert_array_equal(g["xp"], g["x"]); np.testing.assert_array_equal(g["up"], g["u"])       # pre-sharded: the same solve
        np.testing.assert_array_equal(g["cp"], g["costs"])
    np.testing.assert_array_equal(got[0]["u"], got[1]["u"])            # every rank holds the gathered result
    if lockstep:
        assert int(got[0]["n_shard"]) == int(got[1]["n_shard"]) == int(got[0]["n_full"])


def _unconverged_worker(rank, world, port, out_dir):
    for p in (os.path.join(ROOT, "mpc.pytorch_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from mpc import _native, mpc, shard
    from mpc.mpc import LinDx, QuadCost
    from oracle_backend import OracleBackend
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _native.set_backend_for_testing(OracleBackend())
    g = torch.Generator().manual_seed(2)
    T, ns, nc, B = 5, 3, 2, 4
    n = ns + nc
    A = torch.randn(T, B, n, n, generator=g, dtype=torch.float64)
    C = A.transpose(2, 3).matmul(A)
    c = torch.randn(T, B, n, generator=g, dtype=torch.float64)
    F = torch.cat((torch.eye(ns).double() + 0.1 * torch.randn(T - 1, B, ns, ns, generator=g, dtype=torch.float64),
                   torch.randn(T - 1, B, ns, nc, generator=g, dtype=torch.float64)), 3)
    x_init = torch.randn(B, ns, generator=g, dtype=torch.float64)
    # rank 0's block (problems 0, 1) starts AT its optimum (x_init = 0, c = 0 -> u = 0): converged after one
    # step; rank 1's block needs several iterations against its bounds and gets only one
    c[:, :2] = 0
    x_init[:2] = 0
    ctrl = mpc.MPC(ns, nc, T, u_lower=-0.05, u_upper=0.05, lqr_iter=1, verbose=-1, exit_unconverged=True, eps=1e-9)
    raised = 0
    try:
        shard.mpc_forward_sharded(ctrl, x_init, QuadCost(C, c), LinDx(F))
    except mpc.UnconvergedError:
        raised = 1
    np.savez(os.path.join(out_dir, "unc_rank%d.npz" % rank), raised=raised)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_unconverged_error_is_raised_by_every_rank(tmp_path):
    """exit_unconverged=True (mpc/mpc.py:321-324) with one shard converged and the other not: every rank raises
    (the converged one is told through a 1-word all-reduce) instead of one raising and the other waiting forever
    in the all-gather.  mp.spawn(join=True) would hang / time out on the old behaviour."""
    world = 2
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_unconverged_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = [int(np.load(os.path.join(str(tmp_path), "unc_rank%d.npz" % r))["raised"]) for r in range(world)]
    assert got == [1, 1]
