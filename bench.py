#!/usr/bin/env python3
"""bench.py -- LQR problem-steps/s of the MI355X LQR step at BASELINE.json's headline workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--bounded] [--impl {0,1,2,3}]

Workload (configs[3] of BASELINE.json, the one `metric` is quoted on): synthetic random linear
dynamics, n_state=12, n_ctrl=4, T=50, batch=4096 PER GPU, fp32, contiguous time-major tensors
(recipe: SURVEY.md section 8d).  A "step" is one LQRStepFn.forward on that batch -- delta-space
linear term + Riccati sweep + line-searched rollout (mpc/lqr_step.py:277-309 of the reference) --
with every input already resident in HBM.  N > 1: one process per GPU (torch.distributed/RCCL),
each rank owns its own 4096 problems (weak scaling, no data-path collective); the trajectories are
re-assembled with ONE all-gather at the end of the timed region (north_star: "all-gather only to
reassemble trajectories").

Prints ONE JSON line on rank 0: the driver's contract fields + `roofline` (algorithmic bytes /
mean kernel time from HIP events on the launch stream, vs the 8 TB/s HBM peak) + `cpu_baseline`
(the C oracle on the host cores over a bounded sample of the same workload, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd"))
sys.path.insert(0, ROOT)

NS, NC, T_H, B_PER_GPU = 12, 4, 50, 4096
EV_GROUP = 5
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def make_problem(ns, nc, T, B, dtype, device, seed=0, u_scale=0.0, clamp=None, with_f=True):
    """SURVEY.md 8(d) recipe: C = A'A (PSD), c ~ N(0,1), F = [I + 0.2 N/sqrt(ns) | N/sqrt(ns)],
    f = 0.1 N, x_init ~ N(0,1); nominal u = u_scale * N (clamped), nominal x = its rollout."""
    from mpc import util
    from mpc.mpc import LinDx
    g = torch.Generator(device="cpu").manual_seed(seed)
    n = ns + nc
    chunks = []
    for t0 in range(0, T, 10):      # generate in slabs: keeps host memory modest at B = 4096
        A = torch.randn(min(10, T - t0), B, n, n, generator=g, dtype=torch.float32)
        chunks.append(A.transpose(2, 3).matmul(A).to(dtype))
    C = torch.cat(chunks).to(device)
    c = torch.randn(T, B, n, generator=g, dtype=torch.float32).to(dtype).to(device)
    R = torch.eye(ns) + 0.2 * torch.randn(T - 1, B, ns, ns, generator=g) / ns ** 0.5
    S = torch.randn(T - 1, B, ns, nc, generator=g) / ns ** 0.5
    F = torch.cat((R, S), 3).to(dtype).to(device)
    f = (0.1 * torch.randn(T - 1, B, ns, generator=g)).to(dtype).to(device) if with_f else None
    x_init = torch.randn(B, ns, generator=g).to(dtype).to(device)
    u = (u_scale * torch.randn(T, B, nc, generator=g)).to(dtype).to(device)
    if clamp is not None:
        u = u.clamp(-clamp, clamp)
    cur_x = util.get_traj(T, u, x_init, LinDx(F, f))
    return dict(C=C, c=c, F=F, f=f, x_init=x_init, cur_u=u, cur_x=cur_x)


def algorithmic_bytes_per_problem(ns, nc, T, elem=4, with_f=True):
    """SURVEY.md 8(d) / BASELINE.md byte formula: one compulsory pass over
    C, c, F, f, x_init, nominal (x,u) in, new (x,u) out, (cost, du-norm)."""
    n = ns + nc
    return elem * (T * n * n + T * n + (T - 1) * ns * n + ((T - 1) * ns if with_f else 0) + ns + T * n + T * n + 2)


def cpu_baseline(sample_B, bounded, seed=123):
    """The oracle (oracle/, a C restatement of the reference = kind "port") timed on the host cores
    over `sample_B` problems of the same workload.  Only the checker is used here -- as a baseline,
    never as part of the measured GPU path."""
    from oracle import lqr_oracle as O
    p = make_problem(NS, NC, T_H, sample_B, torch.float32, "cuda:0", seed=seed,
                     u_scale=0.3 if bounded else 0.0, clamp=1.0 if bounded else None)
    h = {k: (None if v is None else v.cpu().numpy()) for k, v in p.items()}
    threads = O.max_threads()
    lo, hi = (-1.0, 1.0) if bounded else (None, None)
    args = (h["x_init"], h["C"], h["c"], h["F"], h["f"], h["cur_x"], h["cur_u"], lo, hi)
    O.lqr_step(*args, lockstep=False, nthreads=threads)          # warm
    reps, t0 = 0, time.perf_counter()
    while True:
        O.lqr_step(*args, lockstep=False, nthreads=threads)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or reps >= 2000:
            break
    return dict(value=sample_B * T_H * reps / dt, unit="problem-steps/s", cores=threads, kind="port",
                sample="%d problems x T=%d, %d repetitions in %.1f s, oracle/lqr_oracle.c (C restatement of the reference, per-problem mode), OpenMP over the host cores"
                       % (sample_B, T_H, reps, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=100,
                    help="untimed launches first: the clocks and the Infinity Cache settle over the first ~100 (a step is 0.1 ms)")
    ap.add_argument("--bounded", action="store_true", help="box constraints +-1 (pnqp in the sweep)")
    ap.add_argument("--impl", type=int, default=0, help="0 auto, 1 generic, 2 fused MFMA, 3 DPP 4-problems-per-wave")
    ap.add_argument("--batch", type=int, default=B_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--probe-share", default="", help="diagnostic only: 'C' / 'F' / 'CF' = expand timestep 0 of C / F "
                    "over the horizon (stride 0), which removes that array's HBM traffic without changing the arithmetic")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # (MPC_BENCH_FORCE_DIST=1: take the process-group path at world size 1 too -- a single-GPU box can then
    # exercise the RCCL initialisation and the collectives of the N > 1 run)
    saved_stdout = None
    if world > 1 or os.environ.get("MPC_BENCH_FORCE_DIST"):
        # RCCL writes a version banner to file descriptor 1 when the communicator is created; stdout carries exactly
        # one JSON line, so everything until then goes to stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert args.gpus == world or world == 1, "launch with torch.distributed.run --nproc-per-node N"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from mpc import _native
    from mpc._native import StepOptions
    be = _native.HipBackend()
    _native.load()
    B = args.batch
    p = make_problem(NS, NC, T_H, B, torch.float32, dev, seed=1000 + rank,
                     u_scale=0.3 if args.bounded else 0.0, clamp=1.0 if args.bounded else None)
    opts = StepOptions(u_lower=-1.0, u_upper=1.0) if args.bounded else StepOptions()
    if "C" in args.probe_share:
        p["C"] = p["C"][:1].expand(T_H, -1, -1, -1)
    if "F" in args.probe_share:
        p["F"] = p["F"][:1].expand(T_H - 1, -1, -1, -1)
    impl_used = args.impl if args.impl else (3 if be.impl_supported(NS, NC, torch.float32, 3) else 1)

    # argument structs + output buffers bound once: a timed step is exactly one C-ABI call
    step = be.plan_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], opts, impl=args.impl)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    r = None
    for _ in range(args.warmup):
        r = step()
    gathered = None
    if dist is not None:
        tau = torch.cat((r["new_x"], r["new_u"]), 2)
        gathered = torch.empty((world,) + tuple(tau.shape), dtype=tau.dtype, device=dev)
        dist.all_gather_into_tensor(gathered, tau)
    # HIP events bracket every launch on the stream the kernel runs on (torch's current stream)
    # Event pairs bracket consecutive runs of EV_GROUP launches, back to back, so every launch of the timed
    # region lies inside exactly one pair; an event costs ~3 us of stream time, which neither `value` nor the
    # per-launch figure should pay K times.  kernel_ms = sum of the pairs / K (includes the ~1.5 us launch
    # gaps inside a run: a slight over-estimate of the rocprofv3 kernel duration).
    group = max(1, min(EV_GROUP, args.steps))
    n_ev = (args.steps + group - 1) // group
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i % group == 0:
            ev[i // group][0].record()
        r = step()
        if i % group == group - 1 or i == args.steps - 1:
            ev[i // group][1].record()
    if dist is not None:
        tau = torch.cat((r["new_x"], r["new_u"]), 2)
        dist.all_gather_into_tensor(gathered, tau)
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    if dist is not None:
        tt = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, kern_ms = tt.tolist()

    ok = bool(torch.isfinite(r["costs"]).all().item()) and int(r["status"].max().item()) & 2 == 0
    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = world * B * T_H / (elapsed / args.steps)
        abytes = algorithmic_bytes_per_problem(NS, NC, T_H) * B
        achieved = abytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("bounded" if args.bounded else "unbounded", {}).get(
                    "impl%d" % impl_used)
            except Exception:
                traffic = None
        out = {
            "metric": "LQR problem-steps/sec (batch*T/s), n_state=12 n_ctrl=4 T=50",
            "value": value, "unit": "problem-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic random linear dynamics, n_state=12 n_ctrl=4 T=50 "
                                   "batch=%d per GPU, %s, one LQRStepFn.forward per step"
                                   % (B, "box bounds +-1 (pnqp)" if args.bounded else "unbounded"),
                       "global_batch": world * B, "horizon": T_H, "parallelism": "batch-shard x%d" % world,
                       "kernel": {1: "lqr_step_generic_kernel<float>", 2: "lqr_step_mfma16_kernel", 3: "lqr_step_dpp16_kernel"}[impl_used],
                       "finite": ok},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": abytes, "kernel_ms": kern_ms},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(512, args.bounded)
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
