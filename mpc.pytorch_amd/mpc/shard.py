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
                               nominal_on_dynamics=opts.nominal_on_dynamics,
                               c_symmetric=getattr(opts, "c_symmetric", False))


def all_gather_batch(t, n_batch, dim, group=None):
    """All-gather blocks of unequal size along `dim` (blocks are padded to the largest one on the wire).  General helper for
    small payloads; the trajectories of a step go through GatherSlots below, which needs no packing pass at all."""
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


class GatherSlots:
    """The receive buffer of the ONE all-gather of a step (north_star: "an RCCL all-gather over xGMI only to reassemble
    trajectories"), laid out so that nothing is packed or copied in front of the collective: `world` slots of
    T m (ns + nc) + 3 m reals (m = the largest block), rank r's slot = [ new_x [T,b,ns] | pad | new_u [T,b,nc] | pad |
    costs, full_du_norm, alphas [3,m] ].  The step kernel of rank r WRITES its trajectories into the views `out_x`, `out_u`
    of its own slot, the three per-problem scalars follow with three small copies, and the collective runs in place
    (send buffer = the rank's slot inside the receive buffer, which RCCL recognises).  Round 3 did cat -> expand().contiguous()
    -> movedim().contiguous() -> pad -> all_gather -> cat: four extra passes over the payload and the scalars padded to n
    columns."""

    def __init__(self, T, ns, nc, n_batch, world, rank, dtype, device):
        self.T, self.ns, self.nc, self.world, self.rank = T, ns, nc, world, rank
        self.sizes = [b - a for a, b in (shard_bounds(n_batch, r, world) for r in range(world))]
        self.m = max(max(self.sizes), 1)
        self.xoff, self.uoff, self.soff = 0, T * self.m * ns, T * self.m * (ns + nc)
        self.slot = self.soff + 3 * self.m
        self.buf = torch.empty(world, self.slot, dtype=dtype, device=device)

    def views(self, r):
        """(new_x [T,b,ns], new_u [T,b,nc], scalars [3,b]) of rank r's slot, b = its block size"""
        b, T, ns, nc = self.sizes[r], self.T, self.ns, self.nc
        s = self.buf[r]
        return (s[self.xoff:self.xoff + T * b * ns].view(T, b, ns), s[self.uoff:self.uoff + T * b * nc].view(T, b, nc),
                s[self.soff:self.soff + 3 * self.m].view(3, self.m)[:, :b])

    def gather(self, group=None):
        import torch.distributed as dist
        mine = self.buf[self.rank]
        if dist.get_backend(group) != "nccl":
            mine = mine.clone()             # (gloo in the CPU tests: no promise about aliased send / receive buffers)
        dist.all_gather_into_tensor(self.buf.view(-1), mine, group=group)

    def assembled(self):
        """The whole batch in the reference's layout: new_x [T,B,ns], new_u [T,B,nc], (costs, full_du_norm, alphas) [B] --
        one pass over the payload (a caller that can work on per-rank blocks takes `views(r)` instead: no pass at all)."""
        v = [self.views(r) for r in range(self.world) if self.sizes[r]]
        sc = torch.cat([x[2] for x in v], 1)
        return torch.cat([x[0] for x in v], 1), torch.cat([x[1] for x in v], 1), sc[0], sc[1], sc[2]


def _block_of(n_local, n_batch, rank, world, what):
    """A pre-sharded call hands over the rank's own block; it must be the block `shard_bounds` gives that rank."""
    lo, hi = shard_bounds(n_batch, rank, world)
    if n_local != hi - lo:
        raise ValueError("%s: rank %d of %d owns problems [%d, %d) of %d, got a block of %d" % (what, rank, world, lo, hi, n_batch, n_local))
    return lo, hi


def lqr_step_sharded(x_init, C, c, F, f, cur_x, cur_u, opts, group=None, gather=True, impl=_native.IMPL_AUTO,
                     presharded=False, n_batch=None):
    """One LQR step on this rank's block of the batch; with `gather`, every rank returns the full
    (new_x, new_u, costs, full_du_norm, alphas) after ONE all-gather (GatherSlots: the kernel writes straight into the
    collective's buffer).

    presharded=False: all ranks pass the SAME full-batch tensors (or views of them); only the local block is read.
    presharded=True (what a deployment wants: no rank ever holds the whole batch -- 6.3 GB at config 5 for 0.79 GB of
    work): every tensor, and every tensor-valued option, IS the rank's block, `n_batch` the size of the whole batch (the
    blocks follow shard_bounds; a rank whose block is empty -- n_batch < world -- passes zero-length tensors)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if presharded:
        if n_batch is None:
            raise ValueError("lqr_step_sharded(presharded=True) needs n_batch, the size of the whole batch")
        B = int(n_batch)
        lo, hi = _block_of(C.shape[1], B, rank, world, "lqr_step_sharded")
        local, lopts = (x_init, C, c, F, f, cur_x, cur_u), opts
    else:
        B = C.shape[1]
        lo, hi = shard_bounds(B, rank, world)
        local = (_cut(x_init, lo, hi, 0), _cut(C, lo, hi, 1), _cut(c, lo, hi, 1), _cut(F, lo, hi, 1), _cut(f, lo, hi, 1),
                 _cut(cur_x, lo, hi, 1), _cut(cur_u, lo, hi, 1))
        lopts = shard_options(opts, lo, hi)
    T, ns = C.shape[0], x_init.shape[1]
    nc = C.shape[2] - ns
    slots = None
    kw = {}
    if world > 1 and gather:
        slots = GatherSlots(T, ns, nc, B, world, rank, C.dtype, C.device)
        ox, ou, osc = slots.views(rank)
        kw = dict(out_x=ox, out_u=ou)
    if hi > lo:
        r = _native.backend().lqr_step(*local, lopts, impl=impl, **kw)
    else:               # more ranks than problems: nothing to solve here, the rank still takes part in the collective
        e = lambda *s: torch.empty(*s, dtype=C.dtype, device=C.device)
        r = dict(new_x=e(T, 0, ns), new_u=e(T, 0, nc), costs=e(0), old_costs=e(0), full_du_norm=e(0), alpha_du_norm=e(0),
                 alphas=e(0), qp_iters=torch.zeros(0, dtype=torch.int32, device=C.device),
                 status=torch.zeros(0, dtype=torch.int32, device=C.device))
    if slots is None:
        return r
    if hi > lo:
        osc[0].copy_(r["costs"]); osc[1].copy_(r["full_du_norm"]); osc[2].copy_(r["alphas"])
    slots.gather(group)
    new_x, new_u, costs, du, alphas = slots.assembled()
    return dict(new_x=new_x, new_u=new_u, costs=costs, full_du_norm=du, alphas=alphas, local=r, block=(lo, hi), slots=slots)


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


def mpc_forward_sharded(ctrl, x_init, cost, dx, group=None, lockstep=False, gather=True, presharded=False, n_batch=None):
    """`ctrl(x_init, cost, dx)` (an mpc.MPC) on this rank's block of the batch; with `gather`, every rank returns
    the full (x, u, costs) after one all-gather.  All ranks pass the same full-batch QuadCost / LinDx (or a
    dynamics module); tensor-valued bounds, u_init, u_zero_I of `ctrl` are cut along the batch axis.
    presharded=True: x_init, cost, dx and the tensor-valued fields of `ctrl` ARE the rank's block already (no rank holds the
    whole batch), `n_batch` is the size of the whole batch.
    The returned local block keeps its autograd graph (`local`), the gathered tensors are plain data."""
    import copy
    import torch.distributed as dist
    from .mpc import QuadCost, LinDx, UnconvergedError
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if presharded:
        if n_batch is None:
            raise ValueError("mpc_forward_sharded(presharded=True) needs n_batch, the size of the whole batch")
        B = int(n_batch)
        if B < world:
            # EVERY rank can see this and raises: were it only the ranks whose block is empty, the others would go on into the
            # lock-step all-reduce and the all-gather and wait there for ever
            raise ValueError("mpc_forward_sharded: n_batch %d < world %d leaves rank(s) without a problem to solve" % (B, world))
        lo, hi = _block_of(x_init.shape[0], B, rank, world, "mpc_forward_sharded")
    else:
        B = x_init.shape[0]
        lo, hi = shard_bounds(B, rank, world)
    local = copy.copy(ctrl)
    for name, dim in (("u_lower", 1), ("u_upper", 1), ("u_zero_I", 1), ("u_init", 1), ("prev_ctrl", 0)):
        v = getattr(ctrl, name, None)
        if not presharded and torch.is_tensor(v) and v.dim() > dim and v.shape[dim] == B:
            setattr(local, name, _cut(v, lo, hi, dim))
    if ctrl.n_batch is not None:
        local.n_batch = hi - lo
    local.flag_reducer = lockstep_reducer(group) if (lockstep and world > 1) else None

    def cut_field(t, inner):          # batch axis of a cost / dynamics tensor: the one before its `inner` trailing axes
        if presharded or t is None or not torch.is_tensor(t) or t.dim() <= inner:
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
        x, u, costs = local(x_init if presharded else _cut(x_init, lo, hi, 0), cost, dx)
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
    # the solve's outputs are autograd leaves of its no-op attach step, not buffers this module handed out: one copy of the
    # block into its slot of the collective's buffer (13 MB at the headline shape), then the same in-place all-gather
    slots = GatherSlots(x.shape[0], x.shape[2], u.shape[2], B, world, rank, x.dtype, x.device)
    ox, ou, osc = slots.views(rank)
    ox.copy_(x.detach()); ou.copy_(u.detach()); osc[0].copy_(costs.detach())
    slots.gather(group)
    gx, gu, gc, _, _ = slots.assembled()
    return gx, gu, gc
