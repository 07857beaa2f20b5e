"""Batch sharding of the LQR step over the GPUs of one node (one process per GPU, RCCL over xGMI).

Every tensor of the problem carries the batch as its 2nd axis (1st for x_init) and no arithmetic
mixes problems, so rank r of R simply owns a contiguous block of problems: it solves its block with
the local kernels (no data-path collective) and ONE all-gather reassembles the trajectories
(new_x || new_u, plus the per-problem scalars).  Gradients (dC, dF, ...) stay sharded like their
inputs.  Two of the three batch-global loops of the reference (line search, pnqp -- SURVEY.md section 8e) are
per-problem in the kernels, so shards never need to agree on their trip counts.  The third, the outer iLQR
stop test (max_b full_du_norm < eps, "no problem improved" counter; mpc/mpc.py:271-306), is per shard by
default; `mpc_forward_sharded(..., lockstep=True)` all-reduces its two words per iteration so that every
shard performs exactly the iterations the reference would perform on the whole batch.
"""
import torch

from . import _native


def shard_bounds(n_batch, rank, world):
    """[lo, hi) of the problems rank `rank` of `world` owns: blocks as even as possible, in order."""
    base, extra = divmod(n_batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _cut(t, lo, hi, dim):
    if t is None or not torch.is_tensor(t) or t.numel() == 0:
        return t
    return t.narrow(dim, lo, hi - lo)


def shard_options(opts, lo, hi):
    """The StepOptions of a block of problems: tensor bounds / masks are cut along the batch axis."""
    return _native.StepOptions(u_lower=_cut(opts.u_lower, lo, hi, 1), u_upper=_cut(opts.u_upper, lo, hi, 1),
                               u_zero_I=_cut(opts.u_zero_I, lo, hi, 1), delta_u=opts.delta_u,
                               linesearch_decay=opts.linesearch_decay,
                               max_linesearch_iter=opts.max_linesearch_iter, pnqp_iter=opts.pnqp_iter,
                               true_dynamics=opts.true_dynamics,     # a simulator EnvSpec has no batch axis
                               nominal_on_dynamics=opts.nominal_on_dynamics)


def all_gather_batch(t, n_batch, dim, group=None):
    """All-gather blocks of unequal size along `dim` (blocks are padded to the largest one on the wire)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [b - a for a, b in (shard_bounds(n_batch, r, world) for r in range(world))]
    m = max(sizes)
    t = t.movedim(dim, 0).contiguous()
    if t.shape[0] < m:
        t = torch.cat((t, t.new_zeros((m - t.shape[0],) + tuple(t.shape[1:]))))
    out = t.new_empty((world * m,) + tuple(t.shape[1:]))
    dist.all_gather_into_tensor(out, t, group=group)
    out = torch.cat([out[r * m:r * m + sizes[r]] for r in range(world)])
    return out.movedim(0, dim)


def lqr_step_sharded(x_init, C, c, F, f, cur_x, cur_u, opts, group=None, gather=True, impl=_native.IMPL_AUTO):
    """One LQR step on this rank's block of the batch; with `gather`, every rank returns the full
    (new_x, new_u, costs, full_du_norm, alphas) after one all-gather of the trajectories.

    All ranks pass the SAME full-batch tensors (or views of them); only the local block is read."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = C.shape[1]
    lo, hi = shard_bounds(B, rank, world)
    r = _native.backend().lqr_step(_cut(x_init, lo, hi, 0), _cut(C, lo, hi, 1), _cut(c, lo, hi, 1),
                                   _cut(F, lo, hi, 1), _cut(f, lo, hi, 1), _cut(cur_x, lo, hi, 1),
                                   _cut(cur_u, lo, hi, 1), shard_options(opts, lo, hi), impl=impl)
    if world == 1 or not gather:
        return r
    # ONE collective: trajectories and the per-problem scalars ride in the same buffer
    T = C.shape[0]
    tau = torch.cat((r["new_x"], r["new_u"]), 2)                                       # [T, b, n]
    scal = torch.stack((r["costs"], r["full_du_norm"], r["alphas"]), 1).t().unsqueeze(2)  # [3, b, 1]
    scal = scal.expand(3, hi - lo, tau.shape[2]).contiguous()
    packed = all_gather_batch(torch.cat((tau, scal), 0), B, 1, group)                   # [T+3, B, n]
    ns = r["new_x"].shape[2]
    return dict(new_x=packed[:T, :, :ns], new_u=packed[:T, :, ns:], costs=packed[T, :, 0],
                full_du_norm=packed[T + 1, :, 0], alphas=packed[T + 2, :, 0], local=r, block=(lo, hi))


def lockstep_reducer(group=None):
    """(any_improved, max_du) of this shard -> of the whole batch: one 2-word all-reduce (MAX) per iteration."""
    import torch.distributed as dist

    def reduce(any_improved, max_du):
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.tensor([1.0 if any_improved else 0.0, float("inf") if max_du != max_du else max_du],
                         dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return bool(t[0].item() > 0), float(t[1].item())
    return reduce


def mpc_forward_sharded(ctrl, x_init, cost, dx, group=None, lockstep=False, gather=True):
    """`ctrl(x_init, cost, dx)` (an mpc.MPC) on this rank's block of the batch; with `gather`, every rank returns
    the full (x, u, costs) after one all-gather.  All ranks pass the same full-batch QuadCost / LinDx (or a
    dynamics module); tensor-valued bounds, u_init, u_zero_I of `ctrl` are cut along the batch axis.
    The returned local block keeps its autograd graph (`local`), the gathered tensors are plain data."""
    import copy
    import torch.distributed as dist
    from .mpc import QuadCost, LinDx, UnconvergedError
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = x_init.shape[0]
    lo, hi = shard_bounds(B, rank, world)
    local = copy.copy(ctrl)
    for name, dim in (("u_lower", 1), ("u_upper", 1), ("u_zero_I", 1), ("u_init", 1), ("prev_ctrl", 0)):
        v = getattr(ctrl, name, None)
        if torch.is_tensor(v) and v.dim() > dim and v.shape[dim] == B:
            setattr(local, name, _cut(v, lo, hi, dim))
    if ctrl.n_batch is not None:
        local.n_batch = hi - lo
    local.flag_reducer = lockstep_reducer(group) if (lockstep and world > 1) else None

    def cut_field(t, inner):          # batch axis of a cost / dynamics tensor: the one before its `inner` trailing axes
        if t is None or not torch.is_tensor(t) or t.dim() <= inner:
            return t
        return _cut(t, lo, hi, t.dim() - inner - 1) if t.shape[t.dim() - inner - 1] == B else t
    if isinstance(cost, QuadCost):
        cost = QuadCost(cut_field(cost.C, 2), cut_field(cost.c, 1))
    if isinstance(dx, LinDx):
        dx = LinDx(cut_field(dx.F, 2), cut_field(dx.f, 1))
    # A shard that raises UnconvergedError (exit_unconverged=True, mpc/mpc.py:321-324) must not leave the other
    # ranks blocked in the collective below: every rank learns whether ANY shard failed (one 1-word MAX
    # all-reduce, only when a collective follows) and then all raise, or none does.
    # (only when the solve CAN raise it: exit_unconverged with detach_unconverged, mpc/mpc.py:321-324 -- otherwise the
    # agreement would be a collective and a host synchronisation per forward for nothing.  Any other exception on one
    # rank is a programming error on every rank alike; it is reported to the others the same way so that nobody hangs.)
    can_raise = bool(ctrl.exit_unconverged and ctrl.detach_unconverged)
    err = None
    try:
        x, u, costs = local(_cut(x_init, lo, hi, 0), cost, dx)
    except UnconvergedError as e:
        if world == 1 or not (gather or lockstep):
            raise
        err = e
    except BaseException as e:
        if world > 1 and (gather or lockstep) and can_raise:
            bad = torch.tensor([2.0], device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
        raise
    if world > 1 and (gather or lockstep) and can_raise:
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        bad = torch.tensor([0.0 if err is None else 1.0], device=dev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
        if float(bad.item()) > 1.5 and err is None:
            raise RuntimeError("MPC (sharded): another rank failed inside its solve")
        if float(bad.item()) > 0:
            raise err if err is not None else UnconvergedError(
                "MPC (sharded): another rank's block of problems did not converge (exit_unconverged=True)")
    if world == 1 or not gather:
        return x, u, costs
    T = x.shape[0]
    tau = torch.cat((x.detach(), u.detach()), 2)
    packed = all_gather_batch(torch.cat((tau, costs.detach().view(1, -1, 1).expand(1, hi - lo, tau.shape[2])), 0), B, 1, group)
    ns = x.shape[2]
    return packed[:T, :, :ns], packed[:T, :, ns:], packed[T, :, 0]
