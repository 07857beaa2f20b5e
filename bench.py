#!/usr/bin/env python3
"""bench.py -- LQR problem-steps/s of the MI355X LQR step at BASELINE.json's headline workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--bounded] [--impl {0..5}] [--no-extra]

Workload (configs[3] of BASELINE.json, the one `metric` is quoted on): synthetic random linear
dynamics, n_state=12, n_ctrl=4, T=50, batch=4096 PER GPU, fp32, contiguous time-major tensors
(recipe: SURVEY.md section 8d).  A "step" is one LQRStepFn.forward on that batch -- delta-space
linear term + Riccati sweep + line-searched rollout (mpc/lqr_step.py:277-309 of the reference) --
with every input already resident in HBM.

N > 1: one process per GPU over torch.distributed / RCCL.  Launched either by the driver
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: RANK / WORLD_SIZE in the
environment) or plainly as `python bench.py --gpus N`, in which case this script RE-EXECUTES ITSELF under
torch.distributed.run with N ranks on 127.0.0.1 (and exits non-zero if the node has fewer than N GPUs --
it never falls back to fewer ranks).  Each rank owns its own 4096 problems (weak scaling, no data-path
collective); the trajectories are re-assembled with ONE all-gather at the end of the timed region
(north_star: "all-gather only to reassemble trajectories").

Prints ONE JSON line on rank 0: the driver's contract fields + `roofline` (algorithmic bytes / mean kernel
time from HIP events on the launch stream, vs the 8 TB/s HBM peak) + `cpu_baseline` (the C oracle on the host
cores over a bounded sample of the same workload, rank 0, N=1 only; `reference_probe` = the unmodified
reference timed in the build container, another box, from profiles/ref_cpu_probe.json) + `extra` (N=1 only:
the secondary rows of SURVEY.md 8(d), measured in this same run with the same event method -- box-constrained
step, KKT backward with its own roofline, 5-iteration MPC.forward, config 5 at B=1024 / 8192 with both
fractions, the 10-iteration iLQR solves of configs 2 / 3).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd"))
sys.path.insert(0, ROOT)

NS, NC, T_H, B_PER_GPU = 12, 4, 50, 4096
EV_GROUP = 20            # launches per HIP-event pair (an event pair opens a ~3-10 us gap in the stream)
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_MFMA_PEAK_TF = 157.3  # same guide: v_mfma_f32_16x16x4_f32, dense
SETTLE_LAUNCHES = 200     # the power controller settles over the first ~120 launches of a fresh process: ~17 at boost
                          # clocks (87 us), a dip to 100 us, back to 87-88 us (profiles/r02_kt_durations.json)
KERNEL_NAMES = {1: "lqr_step_generic_kernel<float>", 2: "lqr_step_mfma16_kernel", 3: "lqr_step_dpp16_kernel",
                4: "lqr_step_tiny_kernel", 5: "lqr_step_mfma40_kernel", 8: "lqr_step_dpp16_kernel (padded instantiation)"}


def make_problem(ns, nc, T, B, dtype, device, seed=0, u_scale=0.0, clamp=None, with_f=True, on_device=False):
    """SURVEY.md 8(d) recipe: C = A'A (PSD), c ~ N(0,1), F = [I + 0.2 N/sqrt(ns) | N/sqrt(ns)],
    f = 0.1 N, x_init ~ N(0,1); nominal u = u_scale * N (clamped), nominal x = its rollout.
    on_device: draw the numbers with the device's generator (the big config-5 batches; not reproducible on
    the host, so never used where the CPU oracle has to see the same problem)."""
    from mpc import util
    from mpc.mpc import LinDx
    gdev = device if on_device else "cpu"
    g = torch.Generator(device=gdev).manual_seed(seed)
    n = ns + nc
    kw = dict(generator=g, dtype=torch.float32, device=gdev)
    chunks = []
    for t0 in range(0, T, 10):      # generate in slabs: keeps host memory modest at B = 4096
        A = torch.randn(min(10, T - t0), B, n, n, **kw)
        chunks.append(A.transpose(2, 3).matmul(A).to(dtype).to(device))
    C = torch.cat(chunks)
    c = torch.randn(T, B, n, **kw).to(dtype).to(device)
    R = torch.eye(ns, device=gdev) + 0.2 * torch.randn(T - 1, B, ns, ns, **kw) / ns ** 0.5
    S = torch.randn(T - 1, B, ns, nc, **kw) / ns ** 0.5
    F = torch.cat((R, S), 3).to(dtype).to(device)
    f = (0.1 * torch.randn(T - 1, B, ns, **kw)).to(dtype).to(device) if with_f else None
    x_init = torch.randn(B, ns, **kw).to(dtype).to(device)
    u = (u_scale * torch.randn(T, B, nc, **kw)).to(dtype).to(device)
    if clamp is not None:
        u = u.clamp(-clamp, clamp)
    cur_x = util.get_traj(T, u, x_init, LinDx(F, f))
    return dict(C=C, c=c, F=F, f=f, x_init=x_init, cur_u=u, cur_x=cur_x)


def algorithmic_bytes_per_problem(ns, nc, T, elem=4, with_f=True):
    """SURVEY.md 8(d) / BASELINE.md byte formula: one compulsory pass over
    C, c, F, f, x_init, nominal (x,u) in, new (x,u) out, (cost, du-norm)."""
    n = ns + nc
    return elem * (T * n * n + T * n + (T - 1) * ns * n + ((T - 1) * ns if with_f else 0) + ns + T * n + T * n + 2)


def kkt_algorithmic_bytes_per_problem(ns, nc, T, elem=4, with_f=True):
    """SURVEY.md 8(d), KKT backward: C and dC, F and dF, c and dc, f and df, (x*, u*) and (dl_dx, dl_du), x_init
    and dx_init -- every array read or written once (195,264 B at the headline shape)."""
    n = ns + nc
    return elem * (2 * T * n * n + 2 * (T - 1) * ns * n + 2 * T * n + (2 * (T - 1) * ns if with_f else 0) + 2 * T * n + 2 * ns)


def algorithmic_flops_per_problem_step(ns, nc):
    """2 x MACs of one unconstrained problem-step (17.3 kFLOP at 12/4, 256 kFLOP at 32/8)."""
    n = ns + nc
    mac = (ns * ns * n + n * n * ns + n * ns + nc ** 3 / 3.0 + nc * nc * (ns + 1) + 3 * ns * ns * nc + nc * nc * ns
           + 3 * ns * nc + n * n + nc * ns + ns * n + n * n + 2 * n)
    return 2.0 * mac


def reference_dir():
    """Where the unmodified reference lives on THIS box, or None: $MPC_REFERENCE_DIR, else /root/reference (the build
    container has it, the driver's GPU box does not -- then the port stands alone and says so)."""
    for d in (os.environ.get("MPC_REFERENCE_DIR"), "/root/reference"):
        if d and os.path.isfile(os.path.join(d, "mpc", "lqr_step.py")):
            return d
    return None


def reference_cpu_baseline(host, bounded, reps=3, timeout=600):
    """north_star: "the reference timed on the host cores of the same box (core count stated) in the same run".  The
    UNMODIFIED reference's LQRStep forward (mpc/lqr_step.py:277-309) on `host` -- CPU tensors of a chunk of the very batch
    the kernel was timed on -- in a child interpreter (tools/ref_cpu_child.py: the reference's package and this
    repository's mirror are both called `mpc`), all host threads, median of `reps` calls behind one warm call.
    Returns (row, results) or (None, None) when no reference is on this box."""
    import tempfile
    ref = reference_dir()
    if ref is None:
        return None, None
    tmp = tempfile.mkdtemp(prefix="mpc_refcpu_")
    src, dst = os.path.join(tmp, "in.pt"), os.path.join(tmp, "out.pt")
    z = dict(host)
    z.update(u_lower=-1.0 if bounded else None, u_upper=1.0 if bounded else None, reps=reps)
    torch.save(z, src)
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    try:
        cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_cpu_child.py"), ref, src, dst], env=env,
                            capture_output=True, text=True, timeout=timeout)
        if cp.returncode != 0 or not os.path.exists(dst):
            return {"error": "reference child failed: " + (cp.stderr or "")[-300:]}, None
        o = torch.load(dst)
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}, None
    finally:
        for f in (src, dst):
            if os.path.exists(f):
                os.remove(f)
        os.rmdir(tmp)
    T, B = host["C"].shape[0], host["C"].shape[1]
    row = dict(value=B * T / o["seconds"], unit="problem-steps/s", cores=int(o["cores"]), threads=int(o["threads"]), kind="reference",
               cpu_model=o["cpu_model"], seconds_per_call=o["seconds"], all_seconds=o["all_seconds"],
               sample="%d problems x T=%d of the timed batch, the UNMODIFIED reference's LQRStepFn.forward (%s, torch %s CPU tensors, "
                      "torch.set_num_threads(%d)), median of %d calls behind one warm call" % (B, T, o["reference"], o["torch"], o["threads"], len(o["all_seconds"])))
    return row, o


def reference_parity(o, r, n, bounded):
    """the HIP result of the timed launch against the unmodified reference's own
    float32 run on the same problems, in the same run.  Unbounded: rtol 1e-3 / atol 1e-4, asserted.  Box-constrained: the
    reference's pnqp is batch-global (it iterates until ALL problems of the chunk have converged) and its
    float32 Armijo test is cancellation noise near convergence (CHANGELOG.md 4.5), so its own float32 run deviates from its
    float64 one by more than the tolerance on some problems: reported (share of problems within tolerance), not asserted --
    the asserted check of those rows is `parity`, against the float64 oracle."""
    import numpy as np
    gx, gu = r["new_x"][:, :n].detach().cpu().double().numpy(), r["new_u"][:, :n].detach().cpu().double().numpy()
    ox, ou = o["new_x"].double().numpy(), o["new_u"].double().numpy()
    px = (np.abs(gx - ox) / (1e-4 + 1e-3 * np.abs(ox))).max(axis=(0, 2))
    pu = (np.abs(gu - ou) / (1e-4 + 1e-3 * np.abs(ou))).max(axis=(0, 2))
    within = (np.maximum(px, pu) <= 1.0)
    d = {"problems": int(n), "max_err_over_tol_x": float(px.max()), "max_err_over_tol_u": float(pu.max()),
         "share_within_tol": float(within.mean()), "asserted": not bounded, "tol": "rtol 1e-3 atol 1e-4 (x, u) against the reference's float32 run"}
    d["ok"] = bool(within.all()) if not bounded else bool(within.mean() >= 0.8)
    return d


def cpu_baseline(sample_B, bounded, seed=123, timed=None):
    """The CPU beside the GPU number.  Where the unmodified reference is on this box ($MPC_REFERENCE_DIR or /root/reference):
    kind "reference" -- its LQRStep forward on the first `sample_B` problems of the TIMED batch (`timed` = (problem, result)),
    with in-run parity against the HIP result -- and the port as a second field.  Otherwise kind "port": the oracle
    (oracle/, a C restatement of the reference) on the host cores over `sample_B` problems of the same workload.
    Only the checker is used here -- as a baseline, never as part of the measured GPU path."""
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
    out = dict(value=sample_B * T_H * reps / dt, unit="problem-steps/s", cores=threads, kind="port",
               sample="%d problems x T=%d, %d repetitions in %.1f s, oracle/lqr_oracle.c (C restatement of the reference, per-problem mode), OpenMP over the host cores"
                      % (sample_B, T_H, reps, dt))
    if timed is not None and reference_dir() is not None:
        tp, tr = timed
        n = min(sample_B, int(tp["C"].shape[1]))
        host = {k: (None if tp[k] is None else tp[k].narrow(0 if k == "x_init" else 1, 0, n).detach().cpu().contiguous())
                for k in ("x_init", "C", "c", "F", "f", "cur_x", "cur_u")}
        row, o = reference_cpu_baseline(host, bounded)
        if o is not None:
            row["port"] = out
            row["parity_vs_gpu"] = reference_parity(o, tr, n, bounded)
            out = row
        elif row is not None:
            out["reference_error"] = row["error"]
    else:
        out["reference"] = "not on this box (no $MPC_REFERENCE_DIR, no /root/reference): the port stands in, reference_probe is another box"
    probe = os.path.join(ROOT, "profiles", "ref_cpu_probe.json")
    if os.path.exists(probe):
        try:
            # the UNMODIFIED reference (PyTorch CPU) timed by tools/ref_cpu_probe.py in the build container:
            # another box, another core count -- reported beside the same-box port, never mixed with it
            out["reference_probe"] = json.load(open(probe))
        except Exception:
            pass
    return out


PARITY_SLICE = 64          # problems of the headline batch that go through the oracle
PARITY_ROW = 32            # ... of every secondary row that times a step (config 5: 32 problems x T = 64 x n = 40 in float64)
PARITY_KKT = 16            # ... of every row that times a backward


def _first(t, n, ax):
    import numpy as np
    return None if t is None else t.narrow(ax, 0, n).detach().cpu().numpy().astype(np.float64)


def parity_check(p, r, bounded, n=PARITY_SLICE, lo=None, hi=None, be=None, start=0, converged_nominal=False):
    """the timed problem's own results against the oracle, in the same run.  `n` problems of
    the batch the kernel has just been timed on (from `start`) go through oracle/lqr_oracle.c in float64
    (the checker, never the measured path); tolerance = the one north_star states and the parity tests use (rtol 1e-3 /
    atol 1e-4 on x, u; 5e-4 relative on costs), at every shape (config 5 included: round 4).  The two discontinuities of
    the reference algorithm (tests/test_gpu_fullsize.py: a line-search tie takes the other step size; a box QP's
    minimiser within rounding of a bound counts as clamped or free) are COUNTED, bounded and held to the line search's own
    acceptance rule instead of compared through -- but only once CONFIRMED (round 5): a line-search tie shows in
    the returned alpha; an active-set tie must show in the gains -- the problems in question are solved again by the HIP
    library with K requested (`be`), and the clamped rows of K (exactly zero, mpc/lqr_step.py:142-148) must differ from the
    float64 run's somewhere on the horizon.  A problem out of tolerance with the oracle's own active sets and step size is
    a FAILURE, as is more than max(2, n / 32) ties.
    converged_nominal (round 6, the in-solve row): at a nominal that is already a fixed point every trial's cost equals the nominal's to
    rounding, and the float64 oracle's OWN line search is decided by its rounding there (hundreds of problems of 4096 search several
    trials deep at iteration 4 in float64, tools/iter_probe.py against the oracle) -- a step-size mismatch on a problem whose oracle cost
    moved by less than 1e-5 (1 + |J|) is such a tie and does not count against the cap; it is still held to the acceptance rule."""
    import numpy as np
    from oracle import lqr_oracle as O
    n = min(n, int(r["new_x"].shape[1]) - start)
    if bounded and lo is None:
        lo, hi = -1.0, 1.0

    def cut(t, ax):
        return None if t is None else t.narrow(ax, start, n).detach().cpu().numpy().astype(np.float64)
    # (round 6) the fused float32 kernels of n_state <= 12, n_ctrl <= 4 start every convex box QP from pnqp's own cold start
    # (mpc/pnqp.py:14-19) where the reference passes k of timestep t+1: the oracle makes the same substitution (lqr_oracle.h,
    # qp_cold) -- pnqp's result moves with its start by up to its stopping tolerance of 1e-4, which IS this check's atol
    ns_, nc_ = int(p["x_init"].shape[1]), int(p["C"].shape[-1]) - int(p["x_init"].shape[1])
    qp_cold = bool(lo is not None and p["C"].dtype == torch.float32 and ns_ <= 12 and nc_ <= 4 and not (nc_ == 1 and ns_ <= 6))
    o = O.lqr_step(cut(p["x_init"], 0), cut(p["C"], 1), cut(p["c"], 1), cut(p["F"], 1), cut(p["f"], 1),
                   cut(p["cur_x"], 1), cut(p["cur_u"], 1), lo, hi, lockstep=False, nthreads=O.max_threads(), return_gains=True, qp_cold=qp_cold)
    gx, gu, gc, ga = cut(r["new_x"], 1), cut(r["new_u"], 1), cut(r["costs"], 0), cut(r["alphas"], 0)
    rtol, atol = 1e-3, 1e-4
    px = (np.abs(gx - o["new_x"]) / (atol + rtol * np.abs(o["new_x"]))).max(axis=(0, 2))         # per problem, in units of the tolerance
    pu = (np.abs(gu - o["new_u"]) / (atol + rtol * np.abs(o["new_u"]))).max(axis=(0, 2))
    alpha_tie = ~np.isclose(ga, o["alphas"], rtol=1e-5)
    out_of_tol = (np.maximum(px, pu) > 1.0) & ~alpha_tie
    set_tie = np.zeros(n, bool)
    unexplained = out_of_tol.copy()
    if lo is not None and out_of_tol.any() and be is not None:
        # confirm: the same problems once more with the gains written out; a clamped control is a zero row of K
        from mpc._native import StepOptions
        idx = torch.as_tensor(np.nonzero(out_of_tol)[0] + start, device=r["new_x"].device)
        sub = {k: (None if p[k] is None else p[k].index_select(0 if k == "x_init" else 1, idx).contiguous()) for k in ("x_init", "C", "c", "F", "f", "cur_x", "cur_u")}
        rr = be.lqr_step(sub["x_init"], sub["C"], sub["c"], sub["F"], sub["f"], sub["cur_x"], sub["cur_u"],
                         StepOptions(u_lower=lo, u_upper=hi), want_gains=True)
        gK = rr["K"].detach().cpu().numpy()
        differs = ((gK == 0).all(axis=-1) != (o["K"][:, out_of_tol] == 0).all(axis=-1)).any(axis=(0, 2))
        set_tie[np.nonzero(out_of_tol)[0][differs]] = True
        unexplained = out_of_tol & ~set_tie
    ties = alpha_tie | set_tie
    same = ~(ties | unexplained)
    ec = np.abs(gc - o["costs"]) / np.maximum(1e-12, np.abs(o["costs"]))
    mx = float(px[~ties].max()) if (~ties).any() else 0.0
    mu = float(pu[~ties].max()) if (~ties).any() else 0.0
    mc = float(ec[~ties].max()) if (~ties).any() else 0.0
    # a tie problem took the other branch: its cost is still one the reference could return (not worse than the nominal
    # unless the float64 run is, too)
    old = o["old_costs"]
    tie_ok = bool(np.all((gc[ties] <= old[ties] + 1e-4 * (1 + np.abs(old[ties]))) | (o["costs"][ties] > old[ties] - 1e-4 * (1 + np.abs(old[ties])))))
    flat = alpha_tie & (np.abs(o["costs"] - old) <= 1e-5 * (1 + np.abs(old))) if converged_nominal else np.zeros(n, bool)
    ok = bool(np.isfinite(gx).all() and np.isfinite(gu).all() and mx <= 1.0 and mu <= 1.0 and mc <= 5e-4
              and not unexplained.any() and int((ties & ~flat).sum()) <= max(2, n // 32) and tie_ok)
    return {"ok": ok, "problems": n, "first_problem": int(start), "checker": "oracle/lqr_oracle.c (float64, per-problem mode%s)" % (", box QPs from pnqp's own cold start like the kernel" if qp_cold else ""),
            "tol": "rtol 1e-3 atol 1e-4 (x, u), 5e-4 relative (costs)",
            "max_err_over_tol_x": mx, "max_err_over_tol_u": mu, "cost_rel": mc, "line_search_ties": int(alpha_tie.sum()),
            "active_set_ties": int(set_tie.sum()), "active_set_ties_confirmed_by": "zero rows of K (a second HIP solve of those problems with the gains written out) against the float64 run's",
            "out_of_tolerance_unexplained": int(unexplained.sum()),
            "max_abs_x": float(np.abs(gx - o["new_x"])[:, same].max()) if same.any() else 0.0,
            "max_abs_u": float(np.abs(gu - o["new_u"])[:, same].max()) if same.any() else 0.0}


def parity_slices(p, r, bounded, slices, lo=None, hi=None, be=None, converged_nominal=False):
    """parity_check over several (first problem, count) slices of one batch: one slice -> its dict; several -> ok = all of them,
    the worst figures at the top level, the slices' own dicts beside them."""
    B = int(r["new_x"].shape[1])
    ds = [parity_check(p, r, bounded, n=cnt, lo=lo, hi=hi, be=be, start=max(0, min(s0, B - cnt)), converged_nominal=converged_nominal) for s0, cnt in slices]
    if len(ds) == 1:
        return ds[0]
    out = {"ok": all(d["ok"] for d in ds), "problems": sum(d["problems"] for d in ds), "checker": ds[0]["checker"], "tol": ds[0]["tol"]}
    for key in ("max_err_over_tol_x", "max_err_over_tol_u", "cost_rel"):
        out[key] = max(d[key] for d in ds)
    for key in ("line_search_ties", "active_set_ties", "out_of_tolerance_unexplained"):
        out[key] = sum(d[key] for d in ds)
    out["slices"] = ds
    return out


def kkt_parity_check(p, nx, nu, gx, gu, g, bounded, n=PARITY_KKT):
    """The gradients a timed backward has just written, first `n` problems, against oracle/lqr_oracle.c's restatement of
    LQRStepFn.backward (mpc/lqr_step.py:312-407) in float64 fed the very same (x*, u*, dl_dx, dl_du): every entry within
    2e-4 of its problem's own largest entry of that gradient -- the criterion of tests/test_gpu_fullsize.py."""
    import numpy as np
    from oracle import lqr_oracle as O
    n = min(n, int(nx.shape[1]))
    o = O.kkt_backward(_first(p["C"], n, 1), _first(p["c"], n, 1), _first(p["F"], n, 1), _first(p["f"], n, 1), _first(nx, n, 1), _first(nu, n, 1),
                       _first(gx, n, 1), _first(gu, n, 1), -1.0 if bounded else None, 1.0 if bounded else None, lockstep=False,
                       nthreads=O.max_threads())
    worst, fin = {}, True
    for k in ("dx_init", "dC", "dc", "dF", "df"):
        if g.get(k) is None:
            continue
        bax = 0 if k == "dx_init" else 1
        a = _first(g[k], n, bax)
        fin = fin and bool(np.isfinite(a).all())
        ax = tuple(i for i in range(a.ndim) if i != bax)
        scale = np.maximum(1.0, np.abs(o[k]).max(axis=ax, keepdims=True))
        worst[k] = float((np.abs(a - o[k]) / scale).max())
    ok = fin and all(v < 2e-4 for v in worst.values())
    return {"ok": bool(ok), "problems": n, "checker": "oracle/lqr_oracle.c kkt_backward (float64)",
            "tol": "2e-4 of each problem's largest entry of the gradient", "worst_rel": worst}


PARITY_SOLVE = 16          # problems of a whole iLQR / network solve that are solved again by the float64 checker


def _tol_dict(name, got, ref, rtol, atol):
    import numpy as np
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return float((np.abs(got - ref) / (atol + rtol * np.abs(ref))).max())


def solve_parity(make_ctrl, x0, cost_slices, dyn, out, n=PARITY_SOLVE, rtol=5e-3, atol=5e-3, cost_rtol=1e-4):
    """A WHOLE mpc.MPC solve that has just been timed, certified: its first `n` problems solved again by the same mpc.MPC
    host logic in float64 with the CPU oracle as its kernels (tests/oracle_backend.py on oracle/lqr_oracle.c + oracle/env_oracle.py --
    the checker, installed only for this call) and compared: x, u within rtol / atol, costs relative.  The problems of a batch
    do not interact in these solves (eps = 1e-12, not_improved_lim = 1e6: every iteration runs, the best iterate is kept per problem),
    so a slice solves to what the batch solved to; the float32-vs-float64 floor of such a solve -- the checker's own two precisions
    against each other -- is 1e-4 on u on the problems checked, 7e-3 on the flattest of 1024, 4e-8 on costs
    (tools/solve_parity_floor.py, profiles/r06_solve_parity_floor.log)."""
    import numpy as np
    from mpc import _native
    from mpc.mpc import QuadCost
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_backend import OracleBackend
    ns_, nc_ = int(x0.shape[1]), int(cost_slices[0].shape[-1]) - int(x0.shape[1])
    cold = x0.dtype == torch.float32 and ns_ <= 12 and nc_ <= 4 and not (nc_ == 1 and ns_ <= 6)     # (the kernels' QP start, see parity_check)
    prev = _native.set_backend_for_testing(OracleBackend(qp_cold=cold))
    try:
        ctrl = make_ctrl()
        C64, c64 = (t[:, :n].detach().double().cpu() for t in cost_slices)
        dyn64 = dyn
        from mpc.mpc import LinDx
        if isinstance(dyn, LinDx):          # (linear dynamics: the first n problems' blocks, in float64 on the host)
            dyn64 = LinDx(dyn.F[:, :n].detach().double().cpu(), None if dyn.f is None else dyn.f[:, :n].detach().double().cpu())
        elif isinstance(dyn, torch.nn.Module) and any(True for _ in dyn.parameters()):
            import copy
            dyn64 = copy.deepcopy(dyn).double().cpu()
        with torch.no_grad():
            x, u, c = ctrl(x0[:n].detach().double().cpu(), QuadCost(C64, c64), dyn64)
    finally:
        _native.set_backend_for_testing(prev)
    gx, gu, gc = (t.detach().double().cpu().numpy() for t in (out[0][:, :n], out[1][:, :n], out[2][:n]))
    ex, eu = _tol_dict("x", gx, x.numpy(), rtol, atol), _tol_dict("u", gu, u.numpy(), rtol, atol)
    ec = float((np.abs(gc - c.numpy()) / np.maximum(1e-9, np.abs(c.numpy()))).max())
    return {"ok": bool(np.isfinite(gx).all() and ex <= 1.0 and eu <= 1.0 and ec <= cost_rtol), "problems": int(n),
            "checker": "mpc.MPC (this package's host logic) in float64 on tests/oracle_backend.py = oracle/lqr_oracle.c + oracle/env_oracle.py",
            "tol": "x, u: rtol %g atol %g; costs: %g relative -- a whole solve, float32 kernels against float64" % (rtol, atol, cost_rtol),
            "max_err_over_tol_x": ex, "max_err_over_tol_u": eu, "cost_rel": ec}


def time_launches(fn, steps, warmup, barrier=None):
    """W untimed launches, then exactly K launches bracketed by barrier + synchronize on both sides.
    HIP events bracket consecutive runs of EV_GROUP launches on the stream the kernel runs on (torch's
    current stream), back to back, so every launch of the timed region lies inside exactly one pair; an event
    costs ~3 us of stream time, which neither `value` nor the per-launch figure should pay K times.
    Returns (wall seconds of the K launches, mean ms per launch by the events, last result)."""
    r = None
    for _ in range(warmup):
        r = fn()
    group = max(1, min(EV_GROUP, steps))
    n_ev = (steps + group - 1) // group
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
    (barrier or torch.cuda.synchronize)()
    t0 = time.perf_counter()
    for i in range(steps):
        if i % group == 0:
            ev[i // group][0].record()
        r = fn()
        if i % group == group - 1 or i == steps - 1:
            ev[i // group][1].record()
    return t0, ev, r


def finish_timing(t0, ev, steps, barrier=None):
    (barrier or torch.cuda.synchronize)()
    elapsed = time.perf_counter() - t0
    return elapsed, sum(a.elapsed_time(b) for a, b in ev) / steps


def timed(fn, steps=20, warmup=5):
    t0, ev, r = time_launches(fn, steps, warmup)
    elapsed, kern_ms = finish_timing(t0, ev, steps)
    return elapsed * 1e3 / steps, kern_ms, r


def timed_each(fn, steps=20, warmup=5):
    """Every call bracketed by its own pair of events: -> (median ms, mean ms, max ms, last result).  For the rows whose call is a
    whole solve with host round trips in it: one stalled solve (milliseconds, seen after large blocks went back to the driver)
    moves a 20-solve mean by its full weight, the median says what a solve takes and the mean / max say that it happened."""
    r = None
    for _ in range(warmup):
        r = fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], sum(ts) / len(ts), ts[-1], r


ROW_SETTLE = 150          # launches in front of a secondary row's timed region (see timed_sustained)


def timed_sustained(fn, steps, settle=ROW_SETTLE):
    """A row of `extra` as the headline is measured: every row starts on a GPU that has idled through the host-side set-up of
    its problem (seconds of random numbers), i.e. with ~17 launches at boost clocks and then the power controller's dip
    (+10-15 % per launch) until it settles after ~120 launches (profiles/r02_kt_durations.json).  Rounds 1-3 timed 20
    launches behind 10: the transient itself (the box-constrained step read 204 us in `extra` and 186 us as `--bounded` on
    the same box, same seed).  `settle` launches run first, bracketed by events of their own: the row reports the sustained
    state (`ms`) AND the mean over every launch it made (`ms_all_launches`).  -> (wall ms, ms, ms over all launches, result)"""
    pre = []
    for i0 in range(0, settle, EV_GROUP):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(min(EV_GROUP, settle - i0)):
            fn()
        b.record()
        pre.append((a, b))
    wall, ms, r = timed(fn, steps, 0)
    ms_all = (sum(a.elapsed_time(b) for a, b in pre) + ms * steps) / (settle + steps)
    return wall, ms, ms_all, r


def hbm_roofline(abytes, kern_ms, **more):
    ach = abytes / (kern_ms * 1e-3) / 1e9
    d = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
         "algorithmic_bytes_per_launch": abytes, "kernel_ms": kern_ms}
    d.update(more)
    return d


def extra_rows(be, dev, steps):
    """SURVEY.md 8(d) secondary rows, same run, same event method.  Every row: what was launched, ms per launch
    by HIP events (`ms`), wall ms per launch (`wall_ms`), and a roofline where the bytes are defined."""
    from mpc import mpc, _native
    from mpc._native import StepOptions
    from mpc.mpc import LinDx, QuadCost
    rows = {}
    k = max(10, min(steps, 50))

    def certify(row, fn):
        """Every row that times a step or a backward certifies the launches it has just timed: a slice of their results
        through the oracle, same tolerances as the parity tests; a row that is out makes the whole run exit non-zero."""
        try:
            row["parity"] = fn()
        except Exception as e:
            row["parity"] = {"ok": False, "error": "%s: %s" % (type(e).__name__, e)}
        row["finite"] = bool(row.get("finite", True) and row["parity"]["ok"])
        return row

    def step_row(p, opts, ns, nc, T, B, impl=0, starts=(0,), warm=False, converged_nominal=False):
        """starts: first problems of the slices of PARITY_ROW / len(starts) problems each that go through the oracle (config 5 at
        B = 8192: the first wavefronts, a middle round, the ragged tail).  warm: -> (row, results, row of the SAME step with its box
        QPs started from the k this one left in the workspace -- mpc_lqr_options.qp_start)."""
        plan = be.plan_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], opts, impl=impl)
        settle = ROW_SETTLE if B * T * (ns + nc) ** 2 < 4e8 else 40      # (config 5 at B = 8192: 2.3 ms a launch)
        bounded = opts.u_lower is not None

        def one(fn):
            wall, ms, ms_all, r = timed_sustained(fn, k, settle)
            ok = bool(torch.isfinite(r["costs"]).all().item())
            abytes = algorithmic_bytes_per_problem(ns, nc, T) * B
            row = dict(ms=ms, wall_ms=wall, ms_all_launches=ms_all, settle_launches=settle, problem_steps_per_s=B * T / (ms * 1e-3), finite=ok,
                       roofline=hbm_roofline(abytes, ms, frac_all_launches=abytes / (ms_all * 1e-3) / 1e9 / HBM_PEAK_GBS))
            if bounded:
                row["qp_iterations_per_timestep"] = float(r["qp_iters"].float().mean().item()) / T     # 1 + pnqp iterations (mpc/lqr_step.py:140)
            each = max(8, PARITY_ROW // len(starts))
            return certify(row, lambda: parity_slices(p, r, bounded, [(s0, each) for s0 in starts], lo=opts.u_lower, hi=opts.u_upper, be=be,
                                                      converged_nominal=converged_nominal)), r
        row, r = one(plan)
        if not warm:
            return row, r
        rec = be.qp_record(plan)
        if rec is None:
            return row, r, None
        import copy
        ow = copy.copy(opts)
        ow.qp_start = rec
        roww, _ = one(be.plan_variant(plan, opts=ow))
        roww["qp_start"] = ("mpc_lqr_options.qp_start = the k_t the cold step of this row's problem left in the workspace (mpc_lqr_qp_record; the warm "
                            "step rewrites that record in place): every QP confirms its start in one trip.  What a re-solve at an unchanged nominal "
                            "sees (the re-attach step of MPC.forward, a converged iterate, a receding-horizon re-plan); a fresh nominal is the cold row.")
        return row, r, roww

    def kkt_row(p, r, opts, ns, nc, T, B):
        gx, gu = torch.randn_like(r["new_x"]), torch.randn_like(r["new_u"])
        nx, nu = r["new_x"].clone(), r["new_u"].clone()
        # (pre-bound like the step rows' plan_step: outputs and workspace allocated once -- with an allocation per call the row
        # measured the host's allocator where the kernel is shorter than the call's host time)
        kfn = be.plan_kkt_backward(p["C"], p["c"], p["F"], p["f"], nx, nu, gx, gu, opts)
        if kfn is None:
            kfn = lambda: be.kkt_backward(p["C"], p["c"], p["F"], p["f"], nx, nu, gx, gu, opts)
        wall, ms, ms_all, g = timed_sustained(kfn, k, 100)
        import ctypes
        pf, _keep = be._problem(p["x_init"], p["C"], p["c"], p["F"], p["f"], nx, nu)
        of, _keep2 = opts.to_struct(T, B, nc, p["C"])
        fused = bool(_native.load().mpc_lqr_kkt_fused_supported(ctypes.byref(pf), ctypes.byref(of)))
        kb = kkt_algorithmic_bytes_per_problem(ns, nc, T) * B
        return certify(dict(ms=ms, wall_ms=wall, ms_all_launches=ms_all, settle_launches=100, finite=bool(torch.isfinite(g["dC"]).all().item()),
                    launches=(("mpc_lqr_kkt_fused: ONE launch (sweep + lambda, then rollout + dlambda = V dx + v + all gradients)" if ns <= 12 else
                               "mpc_lqr_kkt_fused: the nested step with lambda along its sweep and dlambda = V dx + v along its rollout, "
                               "then the outer-product kernel (two launches)") if fused
                              else "mpc_lqr_kkt_prepare + mpc_lqr_step (nested solve) + mpc_lqr_kkt_grads"),
                    roofline=hbm_roofline(kb, ms, frac_all_launches=kb / (ms_all * 1e-3) / 1e9 / HBM_PEAK_GBS)),
                       lambda: kkt_parity_check(p, nx, nu, gx, gu, g, opts.u_lower is not None))

    # ---- the headline shape: box-constrained step, KKT backward, 5-iteration MPC.forward ----------------
    for bounded in (False, True):
        key = "bounded" if bounded else "unbounded"
        p = make_problem(NS, NC, T_H, B_PER_GPU, torch.float32, dev, seed=5, u_scale=0.3 if bounded else 0.0,
                         clamp=1.0 if bounded else None)
        # as mpc.MPC calls a step from its second iteration on: the nominal is its own rollout, C has been tested symmetric
        opts = (StepOptions(u_lower=-1.0, u_upper=1.0, nominal_on_dynamics=True, c_symmetric=True) if bounded
                else StepOptions(nominal_on_dynamics=True, c_symmetric=True))
        if bounded:
            row, r, roww = step_row(p, opts, NS, NC, T_H, B_PER_GPU, warm=True)
        else:
            row, r = step_row(p, opts, NS, NC, T_H, B_PER_GPU)
        if not bounded:
            # what ONE MPC.forward sees: five launches on a GPU that has idled (2 s: clocks down, caches cold) -- no settle launches,
            # no warm-up; three such bursts, each between its own pair of events
            planb = be.plan_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], opts)
            bursts = []
            for _ in range(3):
                torch.cuda.synchronize()
                time.sleep(2.0)
                a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _i in range(5):
                    planb()
                b_.record()
                torch.cuda.synchronize()
                bursts.append(a.elapsed_time(b_) / 5)
            ab = algorithmic_bytes_per_problem(NS, NC, T_H) * B_PER_GPU
            med = sorted(bursts)[1]
            rows["headline_burst5_from_idle"] = dict(
                ms=med, ms_each_burst=bursts, launches_per_burst=5, idle_s_before=2.0, roofline=hbm_roofline(ab, med),
                workload="headline step, vouched as mpc.MPC calls it: 5 launches back to back after 2 s of idle, mean per launch, median of 3 bursts "
                         "(the sustained figure needs ~120 launches of run-up, profiles/r02_kt_durations.json; a 5-iteration solve never gets there)")
            del planb
            rowv, _ = step_row(p, StepOptions(), NS, NC, T_H, B_PER_GPU)
            rowv["workload"] = ("headline shape, unbounded, NO promises (a bare LQRStep call): the kernel verifies at every timestep that "
                                "the nominal obeys the dynamics and that C_t is symmetric, and a gated launch of the generic kernel "
                                "behind it re-solves the problems whose C is not (none here)")
            rows["lqr_step_unbounded_verified_nominal"] = rowv
        if bounded:
            row["workload"] = "headline shape, box bounds +-1 (pnqp in the sweep), nominal u ~ 0.3 N clamped"
            rows["lqr_step_bounded"] = row
            if roww is not None:
                roww["workload"] = row["workload"] + "; the box QPs started from the solutions of an earlier step at this nominal"
                rows["lqr_step_bounded_warm"] = roww
            # (round 6) the same step as the FOURTH iteration of a solve sees it: three steps, each from the previous
            # one's result, then this one timed at that nominal -- where rounds 4-5 had a few problems of 4096 search to the last trial
            # (206-216 us: a double rounding of the priced cost) and the QPs are nearly confirmed by their start
            xi, ui = p["cur_x"], p["cur_u"]
            for _ in range(3):
                ri = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], xi, ui, opts)
                xi, ui = ri["new_x"].clone(), ri["new_u"].clone()
            p4 = dict(p, cur_x=xi, cur_u=ui)
            row4, r4 = step_row(p4, opts, NS, NC, T_H, B_PER_GPU, converged_nominal=True)
            row4["workload"] = "headline shape, box bounds +-1: the step of lqr_step_bounded as iteration 4 of the solve that starts there"
            row4["alpha_below_decay"] = int((r4["alphas"] < 0.19).sum().item())       # problems that searched beyond alpha = decay
            rows["lqr_step_bounded_in_solve_iter4"] = row4
            del p4, xi, ui, r4
        rows["kkt_backward_" + key] = kkt_row(p, r, opts, NS, NC, T_H, B_PER_GPU)
        ctrl = mpc.MPC(NS, NC, T_H, u_lower=-1.0 if bounded else None, u_upper=1.0 if bounded else None,
                       lqr_iter=5, verbose=-1, exit_unconverged=False, detach_unconverged=False, backprop=False)
        cost, dx = QuadCost(p["C"], p["c"]), LinDx(p["F"], p["f"])
        wall, ms, out5 = timed(lambda: ctrl(p["x_init"], cost, dx), 12, 25)        # (25 solves = 125 steps in front: the sustained state)

        def mk5(bounded=bounded):
            return mpc.MPC(NS, NC, T_H, u_lower=-1.0 if bounded else None, u_upper=1.0 if bounded else None,
                           lqr_iter=5, verbose=-1, exit_unconverged=False, detach_unconverged=False, backprop=False)
        # (round 6: the whole solve certified like the simulator rows -- its first 16 problems solved again by mpc.MPC on the oracle in
        # float64: gated at 2e-4 in x, u and 1e-4 in costs; five iterations end 1e-5 from that run, costs 4e-6)
        rows["mpc_forward_5iter_" + key] = certify(dict(ms=ms, wall_ms=wall, lqr_iter=5,
                                                        note="whole MPC.forward: initial trajectory kernel + 5 x (step + select_best)"),
                                                   lambda: solve_parity(mk5, p["x_init"], (p["C"], p["c"]), dx, out5, rtol=2e-4, atol=2e-4))
        del p, r, ctrl, cost, dx
    # ---- config 5: n_state=32 n_ctrl=8 T=64, the MFMA tile path; B=1024 is one GPU's share of 8192 over 8 ------
    for B5 in (1024, 8192):
        p = make_problem(32, 8, 64, B5, torch.float32, dev, seed=9, on_device=True)
        row, r = step_row(p, StepOptions(nominal_on_dynamics=True, c_symmetric=True), 32, 8, 64, B5,      # as mpc.MPC calls it
                          starts=(0,) if B5 == 1024 else (0, B5 // 2 - 5, B5 - 11))      # (B = 8192: first wavefronts, a middle round, the tail)
        tf = algorithmic_flops_per_problem_step(32, 8) * B5 * 64 / (row["ms"] * 1e-3) / 1e12
        row["mfma_fp32"] = {"bound": "mfma", "achieved": tf, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                            "frac": tf / FP32_MFMA_PEAK_TF}
        rows["cfg5_step_B%d" % B5] = row
        if B5 == 1024:
            rowv, _ = step_row(p, StepOptions(), 32, 8, 64, B5)
            rowv["workload"] = ("config 5, NO promises (bare LQRStep call): every line-search trial priced from C in the rollout, C tested "
                                "for symmetry in the sweep, gated generic launch behind it")
            rows["cfg5_step_B1024_verified_nominal"] = rowv
            rows["cfg5_kkt_backward_B1024"] = kkt_row(p, r, StepOptions(c_symmetric=True), 32, 8, 64, B5)
            pb = dict(p)
            rowb, _, rowbw = step_row(pb, StepOptions(u_lower=-1.0, u_upper=1.0, nominal_on_dynamics=True, c_symmetric=True), 32, 8, 64, B5, warm=True)   # as mpc.MPC calls it
            rowb["workload"] = "config 5 with box bounds +-1 (pnqp in 8 unknowns in the sweep; line search priced from the sweep's record, no pass over C)"
            rows["cfg5_step_bounded_B1024"] = rowb
            if rowbw is not None:
                rowbw["workload"] = rowb["workload"] + "; the box QPs started from the solutions of an earlier step at this nominal"
                rows["cfg5_step_bounded_B1024_warm"] = rowbw
            # The rows above are back-to-back launches: address translations of C and F stay cached.  An application runs other
            # kernels between two steps; this row puts a 16 us kernel that reads one byte in every 4 KiB page of 800 MB in front
            # of every launch and subtracts it (the 32/8 sweep is latency-bound: what it must not do is wait for TLB misses).
            big = torch.zeros(800 * 1024 * 1024, dtype=torch.uint8, device=dev)
            view = big[::4096]
            _, t_touch, _ = timed(lambda: view.sum(), k, 10)
            plan5 = be.plan_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"],
                                 StepOptions(nominal_on_dynamics=True, c_symmetric=True))
            _, t_cold, _ = timed(lambda: (view.sum(), plan5())[1], k, 10)
            rows["cfg5_step_B1024_cold_translations"] = dict(
                ms=t_cold - t_touch, toucher_ms=t_touch, roofline=hbm_roofline(algorithmic_bytes_per_problem(32, 8, 64) * B5, t_cold - t_touch),
                workload="config 5 step as cfg5_step_B1024, every launch behind a kernel that walks 800 MB of other pages")
            del big, view, plan5
            # whole solves at this shape, like the headline's mpc_forward rows: trajectory kernel + 5 x (step + select_best)
            for bounded5 in (False, True):
                ctrl5 = mpc.MPC(32, 8, 64, u_lower=-1.0 if bounded5 else None, u_upper=1.0 if bounded5 else None, lqr_iter=5,
                                verbose=-1, exit_unconverged=False, detach_unconverged=False, backprop=False)
                cost5, dx5 = QuadCost(p["C"], p["c"]), LinDx(p["F"], p["f"])
                wall5, ms5, out55 = timed(lambda: ctrl5(p["x_init"], cost5, dx5), 12, 12)

                def mk55(bounded5=bounded5):
                    return mpc.MPC(32, 8, 64, u_lower=-1.0 if bounded5 else None, u_upper=1.0 if bounded5 else None, lqr_iter=5,
                                   verbose=-1, exit_unconverged=False, detach_unconverged=False, backprop=False)
                rows["cfg5_mpc_forward_5iter_" + ("bounded" if bounded5 else "unbounded")] = certify(dict(
                    ms=ms5, wall_ms=wall5, lqr_iter=5, note="whole MPC.forward at config 5: initial trajectory kernel + 5 x (step + select_best)"),
                    lambda: solve_parity(mk55, p["x_init"], (p["C"], p["c"]), dx5, out55, n=8, rtol=1e-3, atol=1e-3))
                del ctrl5, cost5, dx5
        del p, r
    # ---- shapes BETWEEN the hand-tuned ones (round 4): the 32/8 kernel's padded instantiation (impl 7 under impl 0), beside the
    # generic kernel every such shape ran on in rounds 1-3 -- the reference's sweep is shape-agnostic (mpc/lqr_step.py:61-158)
    for ns_p, nc_p in ((13, 4), (20, 5), (24, 8)):
        p = make_problem(ns_p, nc_p, T_H, 1024, torch.float32, dev, seed=40 + ns_p)
        row, r_p = step_row(p, StepOptions(nominal_on_dynamics=True, c_symmetric=True), ns_p, nc_p, T_H, 1024)
        rowg, _ = step_row(p, StepOptions(nominal_on_dynamics=True, c_symmetric=True), ns_p, nc_p, T_H, 1024, impl=1)
        if (ns_p, nc_p) == (20, 5):
            # its KKT backward (round 6): the 32/8 kernel's fused backward, padded instantiation (lqr_mfma40_padkkt.o) + the outer-product
            # kernel's general form, where rounds 1-5 took three launches ending in the generic gradient kernel
            rows["pad_kkt_backward_20_5_B1024"] = kkt_row(p, r_p, StepOptions(c_symmetric=True), ns_p, nc_p, T_H, 1024)
        del r_p
        row["workload"] = ("n_state=%d n_ctrl=%d T=%d B=1024, unconstrained: the padded 32/8 kernel (%s gathers); generic_kernel_ms = the same call "
                           "forced onto the generic kernel (rounds 1-3)" % (ns_p, nc_p, T_H, "16-byte" if ns_p % 4 == 0 and nc_p % 4 == 0 else "dword"))
        row["generic_kernel_ms"] = rowg["ms"]
        row["speedup_over_generic"] = rowg["ms"] / row["ms"]
        rows["pad_step_%d_%d_B1024" % (ns_p, nc_p)] = row
        del p
    # ---- shapes UP TO 12/4 (round 6): the 12/4 kernel's padded instantiation (impl 8 under impl 0: dword gathers of the staging DMA pad
    # tau to [x(12); u(4)]), beside the one-problem-per-wavefront kernel (impl 2) every such shape ran on in rounds 1-5
    from mpc._native import IMPL_MFMA16
    for ns_p, nc_p, bnd in ((8, 4, False), (10, 3, False), (12, 2, False), (10, 3, True)):
        p = make_problem(ns_p, nc_p, T_H, B_PER_GPU, torch.float32, dev, seed=60 + ns_p, u_scale=0.3 if bnd else 0.0, clamp=1.0 if bnd else None)
        o_p = (StepOptions(u_lower=-1.0, u_upper=1.0, nominal_on_dynamics=True, c_symmetric=True) if bnd
               else StepOptions(nominal_on_dynamics=True, c_symmetric=True))
        row, r_p = step_row(p, o_p, ns_p, nc_p, T_H, B_PER_GPU)
        rowm, _ = step_row(p, o_p, ns_p, nc_p, T_H, B_PER_GPU, impl=IMPL_MFMA16)
        if (ns_p, nc_p) == (10, 3):
            # the KKT backward of such a shape (round 6): the fused kernel's padded instantiation, lqr_dpp16_padkkt.o -- one launch where
            # rounds 1-5 took three (mpc_lqr_kkt_prepare, the nested step, the generic mpc_lqr_kkt_grads: 0.60 / 0.76 ms)
            rows["pad12_kkt_backward_%d_%d_B%d%s" % (ns_p, nc_p, B_PER_GPU, "_bounded" if bnd else "")] = kkt_row(p, r_p, o_p, ns_p, nc_p, T_H, B_PER_GPU)
        del r_p
        row["workload"] = ("n_state=%d n_ctrl=%d T=%d B=%d, %s: the padded 12/4 kernel (lqr_step_dpp16_kernel, -DMPC_DPP16_PAD); mfma16_kernel_ms = the "
                           "same call forced onto the one-problem-per-wavefront kernel (rounds 1-5)" % (ns_p, nc_p, T_H, B_PER_GPU, "box bounds +-1" if bnd else "unconstrained"))
        row["mfma16_kernel_ms"] = rowm["ms"]
        row["speedup_over_mfma16"] = rowm["ms"] / row["ms"]
        rows["pad12_step_%d_%d_B%d%s" % (ns_p, nc_p, B_PER_GPU, "_bounded" if bnd else "")] = row
        del p
    # ---- float64 (round 5): what every test and gradient check of the reference runs in (tests/test_mpc.py .double()).  n_state <= 12,
    # n_ctrl <= 4 take the one-problem-per-wavefront kernel's float64 instantiation (v_mfma_f64_16x16x4_f64); rounds 1-4: the generic kernel
    for bounded64 in (False, True):
        p = make_problem(NS, NC, T_H, 1024, torch.float64, dev, seed=77, u_scale=0.3 if bounded64 else 0.0, clamp=1.0 if bounded64 else None)
        p["C"] = 0.5 * (p["C"] + p["C"].transpose(2, 3))      # (make_problem multiplies in float32: symmetric to 1e-7 only)
        o64 = StepOptions(u_lower=-1.0, u_upper=1.0) if bounded64 else StepOptions()
        row, _ = step_row(p, o64, NS, NC, T_H, 1024)
        rowg, _ = step_row(p, o64, NS, NC, T_H, 1024, impl=1)
        ab = algorithmic_bytes_per_problem(NS, NC, T_H, elem=8) * 1024
        row["roofline"] = hbm_roofline(ab, row["ms"])
        row["generic_kernel_ms"] = rowg["ms"]
        row["speedup_over_generic"] = rowg["ms"] / row["ms"]
        row["workload"] = ("headline shape in FLOAT64, B=1024%s, a bare LQRStep call: lqr_step_mfma16_f64_kernel; generic_kernel_ms = the same call "
                           "forced onto the generic kernel (rounds 1-4)" % (", box bounds +-1" if bounded64 else ""))
        rows["lqr_step_f64_12_4_B1024" + ("_bounded" if bounded64 else "")] = row
        del p
    torch.cuda.empty_cache()
    # ---- configs 2 / 3: the shipped simulators, whole 10-iteration iLQR solves (L2-resident: latency-bound) ----
    from tools.bench_ilqr_env import problem as env_problem
    for kind, B, T in (("pendulum", 1024, 20), ("cartpole", 4096, 25)):
        dxm, _plain, x0, Q, pp = env_problem(kind, B, T)
        ctrl = mpc.MPC(dxm.n_state, 1, T, u_lower=dxm.lower, u_upper=dxm.upper, lqr_iter=10, verbose=-1,
                       exit_unconverged=False, detach_unconverged=False, linesearch_decay=dxm.linesearch_decay,
                       max_linesearch_iter=dxm.max_linesearch_iter, grad_method=mpc.GradMethods.AUTO_DIFF,
                       eps=1e-12, backprop=False, not_improved_lim=10 ** 6)
        cost = QuadCost(Q, pp)
        # (20 solves behind 6 untimed ones: the first solves after the large blocks of the rows above went back to the driver
        # can stall for milliseconds each -- tools/cfg3_repeat.py --pre --, and 5 timed solves made that a 0.7 / 1.8 ms lottery;
        # round 4 saw the pendulum row at 0.46 ms in two runs and 2.1 / 2.7 ms in two others with the mean of 20
        # -- the row is the MEDIAN of 20 solves timed one by one; mean and maximum beside it.)
        ms, ms_mean, ms_max, out = timed_each(lambda: ctrl(x0, cost, dxm), 20, 6)

        def mk(dxm=dxm, T=T):
            return mpc.MPC(dxm.n_state, 1, T, u_lower=dxm.lower, u_upper=dxm.upper, lqr_iter=10, verbose=-1,
                           exit_unconverged=False, detach_unconverged=False, linesearch_decay=dxm.linesearch_decay,
                           max_linesearch_iter=dxm.max_linesearch_iter, grad_method=mpc.GradMethods.AUTO_DIFF,
                           eps=1e-12, backprop=False, not_improved_lim=10 ** 6)
        rows["cfg%d_ilqr_%s_10iter" % (2 if kind == "pendulum" else 3, kind)] = certify(dict(
            ms=ms, ms_mean=ms_mean, ms_max=ms_max, statistic="median of 20 solves, each between its own events", lqr_iter=10, B=B, T=T,
            ms_per_iteration=ms / 10,
            problem_steps_per_s=B * T * 10 / (ms * 1e-3), mean_cost=float(out[2].mean()),
            note="MPC.forward on mpc.env_dx.%s: the step kernel linearises the simulator and rolls it out itself"
                 % ("PendulumDx" if kind == "pendulum" else "CartpoleDx")),
            # (x, u at 2e-3 on the 16 problems checked: the float32 and the float64 run of the CHECKER itself end 1.2e-4 apart in u
            # on these problems, costs equal to 4e-8 -- 7e-3 in u on the flattest of the first 1024, which is why this is not the
            # step rows' 1e-4: tools/solve_parity_floor.py, profiles/r06_solve_parity_floor.log; round 5 gated at 2e-2)
            lambda: solve_parity(mk, x0, (Q, pp), dxm, out, rtol=2e-3, atol=2e-3))
    # ---- NNDynamics (the reference's default network: one hidden layer of 100 sigmoid units) at the headline shape ----
    from mpc.dynamics import NNDynamics
    torch.manual_seed(0)
    dyn = NNDynamics(NS, NC, [100], activation="sigmoid").to(dev)
    net = dyn.native_net(torch.empty(1, device=dev))
    p = make_problem(NS, NC, T_H, B_PER_GPU, torch.float32, dev, seed=5, u_scale=0.3, clamp=1.0)
    xs, _ = be.mlp_traj_cost(p["x_init"], p["cur_u"], net)
    X, U = xs[:-1].reshape(-1, NS), p["cur_u"][:-1].reshape(-1, NC)
    wall, ms, _ = timed(lambda: be.mlp_traj_cost(p["x_init"], p["cur_u"], net), k, 5)
    # the checker of the network rows: oracle/env_oracle.py (float64 numpy, pinned on the reference's NNDynamics: tests/golden/nn_*.npz)
    import numpy as np
    from oracle import env_oracle as EO
    mlp = EO.Mlp([l.weight.detach().double().cpu().numpy() for l in dyn.fcs], [l.bias.detach().double().cpu().numpy() for l in dyn.fcs],
                 dyn.activation, dyn.passthrough)
    n_nn = PARITY_SOLVE

    def h64(t, ax):
        return t.narrow(ax, 0, n_nn).detach().double().cpu().numpy()

    def nn_parity(pairs, what, **more):
        worst = {k: _tol_dict(k, g, o_, rt, at) for k, (g, o_, rt, at) in pairs.items()}
        fin = all(bool(np.isfinite(np.asarray(g)).all()) for g, _o, _r, _a in pairs.values())
        d = {"ok": bool(fin and all(v <= 1.0 for v in worst.values())), "problems": n_nn, "checker": "oracle/env_oracle.py (float64): " + what,
             "tol": {k: "rtol %g atol %g" % (rt, at) for k, (_g, _o, rt, at) in pairs.items()}, "max_err_over_tol": worst}
        d.update(more)
        return d
    xs_o = EO.traj(EO.MLP, h64(p["x_init"], 0), h64(p["cur_u"], 1), mlp)
    sc = 1.0 + float(np.abs(xs_o).max())
    rows["nn_get_traj"] = certify(dict(ms=ms, wall_ms=wall, problem_steps_per_s=B_PER_GPU * T_H / (ms * 1e-3),
                               workload="util.get_traj through NNDynamics(12, 4, [100], sigmoid), B=%d T=%d: weight packing + "
                                        "one kernel, the network's layers on fp32 MFMA, 16 problems per wavefront" % (B_PER_GPU, T_H)),
                                  lambda: nn_parity({"x": (h64(xs, 1), xs_o, 1e-3, 2e-4 * sc)}, "util.get_traj through the network (mpc/util.py:102-113)"))
    wall, ms, (Fl, fl) = timed(lambda: be.mlp_linearize(net, X, U), k, 5)
    flops = 2.0 * X.shape[0] * (16 * 112 * 16 + 2 * 112 * 16)             # per point: W2 (16x112) G (112x16) + two layer passes, padded tiles
    # (the kernel's Jacobians at ITS OWN trajectory points, first problems: the oracle linearises at the same points)
    Xo, Uo = h64(xs, 1)[:-1].reshape(-1, NS), h64(p["cur_u"], 1)[:-1].reshape(-1, NC)
    Fo, fo = EO.linearize(EO.MLP, Xo, Uo, mlp)
    rows["nn_linearize"] = certify(dict(ms=ms, wall_ms=wall, points=int(X.shape[0]), finite=bool(torch.isfinite(Fl).all().item()),
                                workload="MPC.linearize_dynamics(ANALYTIC) for that network at all (T-1) B points: F [N,12,16], f [N,12]",
                                mfma_fp32={"bound": "mfma", "achieved": flops / (ms * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TF,
                                           "unit": "TFLOP/s", "frac": flops / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF}),
                                   lambda: nn_parity({"F": (h64(Fl.view(T_H - 1, B_PER_GPU, NS, NS + NC), 1).reshape(Fo.shape), Fo, 1e-3, 1e-4),
                                                      "f": (h64(fl.view(T_H - 1, B_PER_GPU, NS), 1).reshape(fo.shape), fo, 1e-3, 1e-4 * sc)},
                                                     "MPC.linearize_dynamics(ANALYTIC) = NNDynamics.grad_input + the affine term (mpc/mpc.py:495-512)"))
    F, f = Fl.view(T_H - 1, B_PER_GPU, NS, NS + NC), fl.view(T_H - 1, B_PER_GPU, NS)
    sw_opts = StepOptions(u_lower=-1.0, u_upper=1.0, max_linesearch_iter=1)
    opts = StepOptions(u_lower=-1.0, u_upper=1.0)
    sw = be.lqr_step(p["x_init"], p["C"], p["c"], F, f, xs, p["cur_u"], sw_opts, want_gains=True)
    wall, ms, rr = timed(lambda: be.mlp_rollout(p["x_init"], p["C"], p["c"], sw["K"], sw["k"], xs, p["cur_u"], sw["old_costs"],
                                                opts, net), k, 5)

    def rollout_parity():
        # the rollout of the first problems again in float64, from the very gains the kernel rolled out with; a line-search tie
        # (the other alpha) is counted, not compared through -- at most 2 of the 16
        nx, nu, cs, _full, al, _tr, _old = EO.rollout_batched(EO.MLP, mlp, h64(p["x_init"], 0), h64(p["C"], 1), h64(p["c"], 1), h64(sw["K"], 1), h64(sw["k"], 1),
                                                              h64(xs, 1), h64(p["cur_u"], 1), -1.0, 1.0, opts.linesearch_decay, opts.max_linesearch_iter)
        same = np.isclose(h64(rr["alphas"], 0), al, rtol=1e-5)
        d = nn_parity({"new_x": (h64(rr["new_x"], 1)[:, same], nx[:, same], 1e-3, 2e-4 * sc), "new_u": (h64(rr["new_u"], 1)[:, same], nu[:, same], 1e-3, 2e-4),
                       "costs": (h64(rr["costs"], 0)[same], cs[same], 1e-3, 1e-6)},
                      "lqr_forward through the network (mpc/lqr_step.py:164-261, module branch :223-225), gains = the timed sweep's", line_search_ties=int((~same).sum()))
        d["ok"] = bool(d["ok"] and (~same).sum() <= 2)
        return d
    rows["nn_rollout_linesearch"] = certify(dict(ms=ms, wall_ms=wall, mean_alpha=float(rr["alphas"].mean()),
                                                 finite=bool(torch.isfinite(rr["costs"]).all().item()),
                                                 workload="lqr_forward with the network as true_dynamics (bounds +-1, up to 10 line-search "
                                                          "passes per problem) after a sweep on the 12/4 kernel"), rollout_parity)
    ctrl = mpc.MPC(NS, NC, T_H, u_lower=-1.0, u_upper=1.0, lqr_iter=5, verbose=-1, exit_unconverged=False,
                   detach_unconverged=False, grad_method=mpc.GradMethods.ANALYTIC, backprop=False, u_init=p["cur_u"].clone())
    cost = QuadCost(p["C"], p["c"])
    with torch.no_grad():
        ms, ms_mean, ms_max, out_nn = timed_each(lambda: ctrl(p["x_init"], cost, dyn), 9, 3)

    def mk_nn():
        return mpc.MPC(NS, NC, T_H, u_lower=-1.0, u_upper=1.0, lqr_iter=5, verbose=-1, exit_unconverged=False, detach_unconverged=False,
                       grad_method=mpc.GradMethods.ANALYTIC, backprop=False, u_init=p["cur_u"][:, :PARITY_SOLVE].detach().double().cpu().clone())
    rows["nn_mpc_forward_5iter"] = certify(dict(ms=ms, ms_mean=ms_mean, ms_max=ms_max, statistic="median of 9 solves, each between its own events", lqr_iter=5,
                                                note="whole MPC.forward on the network: per iteration get_traj + linearisation + sweep + "
                                                     "line-searched rollout kernels; the module called timestep by timestep (this package's "
                                                     "fallback, the reference's only path) takes ~450 ms (tools/nn_bench.py)"),
                                           # (x, u at 2e-2 like the simulator rows: float32 and float64 runs of a 5-iteration solve end 1e-2 apart in u with
                                           # costs equal to 3e-7 -- a flat optimum; the costs are the tight figure)
                                           lambda: solve_parity(mk_nn, p["x_init"], (p["C"], p["c"]), dyn, out_nn, rtol=2e-2, atol=2e-2))
    return rows


def attach_traffic(rows):
    """HBM bytes per launch from the tracked rocprofv3 counter summaries (profiles/r03_prof_*.json: separate --pmc FETCH_SIZE /
    WRITE_SIZE passes, (2 x FETCH + WRITE) x 1024 per MI355X_MICROARCH.md), per kernel; a row that is ONE kernel launch gets
    `roofline.traffic` and `roofline.traffic_over_algorithmic`.  Counter runs are separate runs of the same commands
    (tools/prof_any.sh), never mixed with the timed ones."""
    def load(*tags):
        # (the newest round's summary of that call kind: tools/prof_one.py profiles ONE kind of call per file since round 4)
        for tag in tags:
            try:
                return json.load(open(os.path.join(ROOT, "profiles", tag + ".json")))["pmc_avg_per_dispatch"]
            except Exception:
                continue
        return {}
    kkt, kktb = load("r06_prof_kkt", "r05_prof_kkt", "r04_prof_kkt", "r03_prof_kkt"), load("r04_prof_kkt_bounded", "r03_prof_kkt")
    cfg5, cfg5b = load("r06_prof_cfg5_kkt", "r05_prof_cfg5_kkt", "r04_prof_cfg5_kkt"), load("r06_prof_cfg5_bounded", "r05_prof_cfg5_bounded", "r04_prof_cfg5_bounded")
    bnd = load("r06_prof_bounded", "r05_prof_bounded", "r04_prof_bounded", "r03_prof_bounded")
    c5, p12, p12b = load("r06_prof_cfg5", "r05_prof_cfg5"), load("r06_prof_pad12_10_3"), load("r06_prof_pad12_10_3_bounded")
    # (config 5's backward is two launches: the fused kernel + the outer products; their counters add up)
    k5, o5 = cfg5.get("lqr_kkt_fused_mfma40_kernel<0>"), cfg5.get("kkt_outer_kernel")
    kkt5 = ({"hbm_bytes_per_dispatch": k5["hbm_bytes_per_dispatch"] + o5["hbm_bytes_per_dispatch"]}
            if k5 and o5 and "hbm_bytes_per_dispatch" in k5 and "hbm_bytes_per_dispatch" in o5 else None)
    pick = {"lqr_step_bounded": bnd.get("lqr_step_dpp16_kernel<2>"),
            "kkt_backward_unbounded": kkt.get("lqr_kkt_fused_dpp16_kernel<false>"),
            "kkt_backward_bounded": kktb.get("lqr_kkt_fused_dpp16_kernel<true>"),
            "cfg5_kkt_backward_B1024": kkt5,
            "cfg5_step_B1024": c5.get("lqr_step_mfma40_kernel<0>"),
            "pad12_step_10_3_B4096": p12.get("lqr_step_dpp16_kernel<0>"),
            "pad12_step_10_3_B4096_bounded": p12b.get("lqr_step_dpp16_kernel<2>"),
            "cfg5_step_bounded_B1024": cfg5b.get("lqr_step_mfma40_kernel<2>")}
    for key, v in pick.items():
        if v and key in rows and "roofline" in rows[key] and "hbm_bytes_per_dispatch" in v:
            r = rows[key]["roofline"]
            r["traffic"] = v["hbm_bytes_per_dispatch"]
            r["traffic_over_algorithmic"] = v["hbm_bytes_per_dispatch"] / r["algorithmic_bytes_per_launch"]
    return rows


def dist_row(be, dist, dev, world, rank, ns, nc, T, B_total, seed, steps, what):
    """One row of the N > 1 line: `B_total` problems sharded over the ranks (mpc.shard.shard_bounds: contiguous blocks, no
    data-path collective), every rank's kernel writing straight into its slot of the all-gather's receive buffer
    (mpc.shard.GatherSlots), ONE in-place all_gather_into_tensor inside the timed region, barrier + synchronize on both sides,
    MAX over ranks; 16 problems of every rank's block through the oracle (MIN over ranks).  Collective on every rank; the
    returned dict matters on rank 0."""
    from mpc import shard
    from mpc._native import StepOptions
    lo, hi = shard.shard_bounds(B_total, rank, world)
    b = hi - lo
    p = make_problem(ns, nc, T, b, torch.float32, dev, seed=seed + rank, on_device=(ns > 16))
    slots = shard.GatherSlots(T, ns, nc, B_total, world, rank, torch.float32, dev)
    ox, ou, osc = slots.views(rank)
    step = be.plan_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"],
                        StepOptions(nominal_on_dynamics=True, c_symmetric=True), out_x=ox, out_u=ou)

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()
    r = None
    for _ in range(ROW_SETTLE if ns <= 16 else 40):          # the sustained state, as for every other row (timed_sustained)
        r = step()
    slots.gather()
    t0, ev, r = time_launches(step, steps, 0, barrier)
    osc[0].copy_(r["costs"]); osc[1].copy_(r["full_du_norm"]); osc[2].copy_(r["alphas"])
    slots.gather()
    elapsed, kern_ms = finish_timing(t0, ev, steps, barrier)
    tt = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed, kern_ms = tt.tolist()
    # what arrived: every rank's block of the gathered buffer against the checksum that rank computed of its own block
    mine = torch.stack((ox.double().sum(), ou.double().sum(), osc.double().sum()))
    sums = torch.empty(world, 3, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(sums, mine)
    got = torch.stack([torch.stack([v.double().sum() for v in slots.views(q)]) for q in range(world)])
    arrived = bool(torch.allclose(got, sums, rtol=1e-12, atol=0.0)) and bool(torch.isfinite(got).all())
    try:
        par = parity_check(p, r, False, n=16)
    except Exception as e:
        par = {"ok": False, "error": "%s: %s" % (type(e).__name__, e)}
    okt = torch.tensor([1.0 if (par["ok"] and arrived) else 0.0], device=dev)
    dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    ms = elapsed * 1e3 / steps
    return dict(workload=what, global_batch=B_total, per_rank_batch=[shard.shard_bounds(B_total, q, world)[1] - shard.shard_bounds(B_total, q, world)[0] for q in range(world)],
                ms_per_step=ms, kernel_ms=kern_ms, value=B_total * T / (ms * 1e-3), unit="problem-steps/s",
                roofline=hbm_roofline(algorithmic_bytes_per_problem(ns, nc, T) * b, kern_ms),
                collective="one in-place all_gather_into_tensor of the slots [world, T m n + 3 m] (RCCL) inside the timed region",
                gathered_blocks_match=arrived, parity_rank0=par, parity_all_ranks_ok=bool(okt.item() > 0), finite=bool(okt.item() > 0))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: run N ranks of this very script under torch.distributed.run.
    Refuses (non-zero exit) when the node cannot give N GPUs -- never a silent fall-back to fewer ranks."""
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d asked for, %d visible on this node -- refusing to run fewer ranks\n"
                         % (args.gpus, have))
        sys.exit(3)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MPC_BENCH_SPAWNED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


CONTRACT_LINE_MAX = 4096


def _sig(v, digits=5):
    """Numbers of the contract line: 5 significant digits (the full record keeps every digit)."""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        return float("%.*g" % (digits, v)) if v == v and abs(v) != float("inf") else None
    if isinstance(v, dict):
        return {k: _sig(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_sig(x, digits) for x in v]
    return v


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def contract_line(out):
    """The ONE stdout line of the driver's contract, <= CONTRACT_LINE_MAX bytes whatever the run recorded: the contract keys,
    `config` / `roofline` / `cpu_baseline` / `parity` without their prose, and `extra_digest` = {row: [ms, roofline frac,
    parity ok]} of the secondary rows.  Everything dropped here is in the full record (stderr, bench_full.json)."""
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    cfg = out.get("config", {})
    line["config"] = _pick(cfg, ("workload", "global_batch", "horizon", "parallelism", "kernel", "settle_launches", "finite", "ranks_seen"))
    if isinstance(cfg.get("problem_sets"), str):
        line["config"]["problem_sets"] = int(cfg["problem_sets"].split()[0]) if cfg["problem_sets"][0].isdigit() else 1
    if cfg.get("collective"):
        line["config"]["collective"] = "all_gather_into_tensor"
    rf = out.get("roofline", {})
    line["roofline"] = _pick(rf, ("bound", "achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch", "kernel_ms", "traffic",
                                  "frac_all_launches", "launches_all", "repeat_kernel_ms"))
    if rf.get("traffic_source"):
        line["roofline"]["traffic_source"] = rf["traffic_source"].split(" ")[0]
    if isinstance(rf.get("same_set"), dict):
        line["roofline"]["same_set"] = _pick(rf["same_set"], ("kernel_ms", "frac"))
    if "cpu_baseline" in out:
        cb = out["cpu_baseline"]
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind"))
        line["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:160]
    if "parity" in out:
        line["parity"] = _pick(out["parity"], ("ok", "problems", "tol", "max_err_over_tol_x", "max_err_over_tol_u", "cost_rel",
                                               "line_search_ties", "active_set_ties", "error"))
    ex = out.get("extra")
    if isinstance(ex, dict):
        dig = {}
        for k, v in ex.items():
            if not isinstance(v, dict):
                dig[k] = str(v)[:80]
                continue
            par = v.get("parity")
            pok = par.get("ok") if isinstance(par, dict) else v.get("parity_all_ranks_ok")       # (N > 1 rows: MIN over ranks)
            dig[k] = [v.get("ms", v.get("ms_per_step")), (v.get("roofline") or {}).get("frac"), pok]
            if "value" in v:                      # N > 1 rows: the whole-job rate of that row too
                dig[k].append(v["value"])
        line["extra_digest"] = dig
        line["extra_digest_columns"] = ["ms", "roofline_frac", "parity_ok"]
    if out.get("extra_rows_out_of_tolerance"):
        line["extra_rows_out_of_tolerance"] = out["extra_rows_out_of_tolerance"]
    line["full_record"] = "stderr + bench_full.json"
    line = _sig(line)
    s = json.dumps(line, separators=(",", ":"))
    # belt and braces: shed the optional parts, least important first, until the line fits
    for drop in ("extra_digest_columns", "extra_digest", "parity"):
        if len(s) <= CONTRACT_LINE_MAX:
            break
        if drop == "extra_digest" and isinstance(line.get(drop), dict):
            line[drop] = {k: v[0] if isinstance(v, list) else None for k, v in line[drop].items()}      # ms only
            s = json.dumps(line, separators=(",", ":"))
            if len(s) <= CONTRACT_LINE_MAX:
                break
        line.pop(drop, None)
        s = json.dumps(line, separators=(",", ":"))
    assert len(s) <= CONTRACT_LINE_MAX, len(s)
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=100,
                    help="untimed launches of the step immediately before the timed region")
    ap.add_argument("--bounded", action="store_true", help="box constraints +-1 (pnqp in the sweep)")
    ap.add_argument("--impl", type=int, default=0, help="0 auto, 1 generic, 2 fused MFMA, 3 DPP 4-problems-per-wave")
    ap.add_argument("--batch", type=int, default=B_PER_GPU)
    ap.add_argument("--seed", type=int, default=1000, help="seed of the synthetic problem (rank r uses seed + r)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary rows (`extra`)")
    ap.add_argument("--one-set", action="store_true", help="re-launch ONE problem set (rounds 1-4) instead of alternating between two")
    ap.add_argument("--verify-nominal", action="store_true",
                    help="do not vouch for the nominal: the kernel verifies at every timestep that current_x is the rollout "
                         "of current_u (what a bare LQRStep(...) call with an arbitrary nominal gets)")
    ap.add_argument("--probe-share", default="", help="diagnostic only: 'C' / 'F' / 'CF' = expand timestep 0 of C / F "
                    "over the horizon (stride 0), which removes that array's HBM traffic without changing the arithmetic")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if not launched and args.gpus > 1:
        respawn_under_torchrun(args)          # does not return
    if launched and world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE) -- refusing\n" % (args.gpus, world))
        sys.exit(3)
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
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if torch.cuda.device_count() <= local:
            sys.stderr.write("bench.py: rank %d has no GPU %d on this node\n" % (rank, local))
            sys.exit(3)
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        assert dist.get_world_size() == world
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from mpc import _native
    from mpc._native import StepOptions
    be = _native.HipBackend()
    _native.load()
    B = args.batch
    # TWO distinct problem sets (round 5): the timed launches alternate between them, so that no launch finds anything of
    # its own inputs in the 256 MB Infinity Cache from the launch before (a set is 413 MB; re-launching ONE set leaves the
    # late-t blocks of F and the nominal there for the next sweep).  `roofline.same_set` is the re-launch figure of rounds 1-4.
    psets = [make_problem(NS, NC, T_H, B, torch.float32, dev, seed=args.seed + rank + 7919 * i,
                          u_scale=0.3 if args.bounded else 0.0, clamp=1.0 if args.bounded else None) for i in range(1 if args.one_set else 2)]
    p = psets[0]
    # the nominal IS util.get_traj of the nominal controls (make_problem), as MPC.forward hands it to every step
    # (mpc/mpc.py:251): the step is told so (MPC_OPT_NOMINAL_ON_DYNAMICS), like mpc.MPC does; --verify-nominal times the
    # general entry, which checks the premise at every timestep
    # and C = A'A is symmetric (MPC_OPT_C_SYMMETRIC): mpc.MPC makes that promise from its second iteration on, after the
    # first step of the solve has tested C on the device; --verify-nominal drops both promises (a bare LQRStep call)
    vouch = not args.verify_nominal
    opts = (StepOptions(u_lower=-1.0, u_upper=1.0, nominal_on_dynamics=vouch, c_symmetric=vouch) if args.bounded
            else StepOptions(nominal_on_dynamics=vouch, c_symmetric=vouch))
    for q in psets:
        if "C" in args.probe_share:
            q["C"] = q["C"][:1].expand(T_H, -1, -1, -1)
        if "F" in args.probe_share:
            q["F"] = q["F"][:1].expand(T_H - 1, -1, -1, -1)
    impl_used = args.impl if args.impl else (3 if be.impl_supported(NS, NC, torch.float32, 3) else 1)

    # N > 1: the kernel writes its trajectories straight into this rank's slot of the all-gather's receive buffer
    slots = None
    out_kw = {}
    if dist is not None:
        from mpc import shard
        slots = shard.GatherSlots(T_H, NS, NC, world * B, world, rank, torch.float32, dev)
        out_kw = dict(out_x=slots.views(rank)[0], out_u=slots.views(rank)[1])
    # argument structs + output buffers bound once: a timed step is exactly one C-ABI call
    plans = [be.plan_step(q["x_init"], q["C"], q["c"], q["F"], q["f"], q["cur_x"], q["cur_u"], opts, impl=args.impl, **out_kw) for q in psets]
    turn = [0]

    def step():
        # (set 0, set 1, set 0, ...: one C-ABI call per step either way)
        i = turn[0]
        turn[0] = (i + 1) % len(plans)
        return plans[i]()
    step.outputs = plans[0].outputs

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up launches (not warm-up steps of the contract, reported as `settle_launches`): a fresh process sees ~17
    # launches at boost clocks, then the power controller's dip (+15 % per launch), then the sustained state from
    # launch ~120 on; W = 5 of the driver's default call alone would time the transient
    settle = max(0, SETTLE_LAUNCHES - args.warmup)
    # (the set-up and warm-up launches are bracketed by HIP events as well, EV_GROUP to a pair: `roofline.frac_all_launches`
    # is the figure over EVERY launch of this process, power-controller dip included -- what rocprofv3's per-kernel average
    # of this command reports; `frac` is the sustained state, the timed region)
    pre_ev = []
    r = None
    for n_pre in (settle, args.warmup):
        for i0 in range(0, n_pre, EV_GROUP):
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(min(EV_GROUP, n_pre - i0)):
                r = step()
            b_.record()
            pre_ev.append((a, b_))
    if r is None:
        r = step.outputs
    if dist is not None:
        slots.gather()                       # (communicator and buffers warm before the timed region)
    t0, ev, r = time_launches(step, args.steps, 0, barrier)
    if dist is not None:
        osc = slots.views(rank)[2]
        osc[0].copy_(r["costs"]); osc[1].copy_(r["full_du_norm"]); osc[2].copy_(r["alphas"])
        slots.gather()
    elapsed, kern_ms = finish_timing(t0, ev, args.steps, barrier)
    p = psets[(turn[0] - 1) % len(plans)]          # the set of the LAST timed launch: its results are what `r` holds and what is certified
    n_all = settle + args.warmup + args.steps
    same_set = None
    if dist is None and len(plans) > 1:
        # the figure of rounds 1-4 beside it: the same number of launches on ONE set, back to back (outside the timed region)
        _w, ms_same, _r = timed(plans[0], args.steps, 0)
        same_set = {"kernel_ms": ms_same, "frac": algorithmic_bytes_per_problem(NS, NC, T_H) * B / (ms_same * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "note": "one problem set re-launched back to back (how rounds 1-4 timed the headline): the previous launch's rollout leaves "
                            "late-t blocks in the Infinity Cache for the next launch's sweep"}
        r = plans[(turn[0] - 1) % len(plans)]()      # (the certified results again: plan outputs are overwritten by every call)
        torch.cuda.synchronize()
    # (round 5) ... and the timed region's own launches once more, behind it: `value` is the K steps of the contract whatever happened in
    # them; a box that stalled for milliseconds inside (seen once in ten runs: 0.79 ms per step where every other measure of the
    # run said 0.09) shows as `repeat_kernel_ms` far below `kernel_ms`
    repeat_ms = None
    if dist is None:
        _w, repeat_ms, _r = timed(step, args.steps, 0)
        r = plans[(turn[0] - 1) % len(plans)]() if len(plans) > 1 else step()
        p = psets[(turn[0] - 1) % len(plans)]
        torch.cuda.synchronize()
    kern_ms_all = (sum(a.elapsed_time(b_) for a, b_ in pre_ev) + kern_ms * args.steps) / max(1, n_all)
    ranks_seen = 1
    if dist is not None:
        tt = torch.tensor([elapsed, kern_ms, kern_ms_all], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, kern_ms, kern_ms_all = tt.tolist()
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())

    exit_code = 0
    ok = bool(torch.isfinite(r["costs"]).all().item()) and not bool((r["status"] & 2).any().item())
    if dist is not None:
        okt = torch.tensor([1.0 if ok else 0.0], device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item() > 0)
        # the gathered buffer holds every rank's trajectories: every block must be what its rank computed (checksums of the
        # blocks as their owners see them, gathered beside them) -- and ranks_seen ranks took part in the collectives
        ox, ou, osc = slots.views(rank)
        mine = torch.stack((ox.double().sum(), ou.double().sum(), osc.double().sum()))
        sums = torch.empty(world, 3, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(sums, mine)
        got = torch.stack([torch.stack([v.double().sum() for v in slots.views(q)]) for q in range(world)])
        arrived = bool(torch.allclose(got, sums, rtol=1e-12, atol=0.0)) and bool(torch.isfinite(got).all())
        # ... and the in-place collective (send buffer = this rank's slot INSIDE the receive buffer) against a plain out-of-place
        # all_gather of a copy of that slot (the in-place RCCL path is exercised on a GPU box only)
        plain = [torch.empty_like(slots.buf[rank]) for _ in range(world)]
        dist.all_gather(plain, slots.buf[rank].clone())
        arrived = arrived and all(bool(torch.equal(plain[q], slots.buf[q])) for q in range(world))
        # + 16 problems of every rank's block through the oracle
        try:
            par_rank = parity_check(p, r, args.bounded, n=16, be=be)
        except Exception as e:
            par_rank = {"ok": False, "error": "%s: %s" % (type(e).__name__, e)}
        okt = torch.tensor([1.0 if (par_rank["ok"] and arrived and ranks_seen == world) else 0.0], device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = ok and bool(okt.item() > 0)
    dist_extra = None
    if dist is not None and not args.no_extra:
        # SURVEY.md 8(d).4 / BASELINE configs[3], configs[4]: the STRONG-scaling case (4096 problems over the ranks) and
        # config 5 (8192 problems of 32/8, T = 64 over the ranks) beside the weak-scaling headline
        k_d = max(10, min(args.steps, 50))
        dist_extra = {
            "strong_scaling": dist_row(be, dist, dev, world, rank, NS, NC, T_H, B_PER_GPU, 3000, k_d,
                                       "headline shape, %d problems in all (%d per rank): strong scaling, BASELINE configs[3]" % (B_PER_GPU, B_PER_GPU // world)),
            "cfg5": dist_row(be, dist, dev, world, rank, 32, 8, 64, 8192, 4000, max(5, k_d // 4),
                             "config 5 (n_state=32 n_ctrl=8 T=64), 8192 problems in all (%d per rank): BASELINE configs[4]" % (8192 // world)),
        }
    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = world * B * T_H / (elapsed / args.steps)
        abytes = algorithmic_bytes_per_problem(NS, NC, T_H) * B
        traffic, traffic_source = None, None
        # (the per-kernel counter summary of tools/prof_any.sh, newest round first; "headline_alt" = two alternating problem sets,
        # like this run's timed region)
        for rnd, kindname in (("r06", "bounded" if args.bounded else ("headline" if args.one_set else "headline_alt")),
                              ("r05", "bounded" if args.bounded else ("headline" if args.one_set else "headline_alt")),
                              ("r05", "bounded" if args.bounded else "headline"), ("r04", "bounded" if args.bounded else "headline"),
                              ("r03", "bounded" if args.bounded else "headline")):
            try:
                tfile = "profiles/%s_prof_%s.json" % (rnd, kindname)
                pm = json.load(open(os.path.join(ROOT, tfile)))["pmc_avg_per_dispatch"]
                traffic = pm["lqr_step_dpp16_kernel<%d>" % (2 if args.bounded else 0)]["hbm_bytes_per_dispatch"] if impl_used == 3 else None
                # NOT a counter of this run: rocprofv3 --pmc passes of the same call kind (tools/prof_one.py), separate runs
                traffic_source = tfile + " (pmc_avg_per_dispatch: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/prof_one.py, not this run)"
                break
            except Exception:
                traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if traffic is None and os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("bounded" if args.bounded else "unbounded", {}).get(
                    "impl%d" % impl_used)
                traffic_source = "profiles/hbm_traffic.json (round-2 counter passes, not this run)" if traffic is not None else None
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
                       "kernel": KERNEL_NAMES.get(impl_used, "impl %d" % impl_used),
                       "nominal": "util.get_traj of the nominal controls" + (", flagged on-dynamics and C flagged symmetric as mpc.MPC flags them from its second iteration on"
                                                                                 if vouch else ", nominal and symmetry of C verified by the kernel at every timestep"),
                       "settle_launches": settle, "finite": ok,
                       "problem_sets": "%d distinct sets of %d problems, launches alternate between them" % (len(plans), B) if len(plans) > 1 else "one set, re-launched",
                       "launcher": ("torch.distributed.run (self-spawned by bench.py)" if os.environ.get("MPC_BENCH_SPAWNED")
                                    else "torch.distributed.run") if launched else "single process",
                       "collective": None if dist is None else ("one in-place all_gather_into_tensor (RCCL) inside the timed region: every rank's kernel writes "
                                                                "new_x, new_u into its slot of the receive buffer (mpc.shard.GatherSlots), scalars in 3 B words; "
                                                                "checked bit for bit against an out-of-place all_gather after the timed region"),
                       "ranks_seen": ranks_seen},
            "roofline": hbm_roofline(abytes, kern_ms, traffic=traffic, traffic_source=traffic_source, kernel_ms_all_launches=kern_ms_all,
                                     frac_all_launches=abytes / (kern_ms_all * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     launches_all=n_all, problem_sets=len(plans), same_set=same_set, repeat_kernel_ms=repeat_ms),
        }
        if dist is not None:
            out["parity"] = dict(par_rank, scope="rank 0's first 16 problems; every rank checks its own 16 and the run's `finite` is the MIN over ranks")
            if dist_extra is not None:
                out["extra"] = dist_extra
                if not all(v.get("finite", True) for v in dist_extra.values()):
                    out["config"]["finite"] = ok = False
        if dist is None:
            # self-certification: the results of the launches just timed, against the oracle
            try:
                par = parity_check(p, r, args.bounded, be=be)
            except Exception as e:
                par = {"ok": False, "error": "%s: %s" % (type(e).__name__, e)}
            out["parity"] = par
            if not par["ok"]:
                out["config"]["finite"] = ok = False
        if dist is None and not args.no_extra:
            try:
                out["extra"] = attach_traffic(extra_rows(be, dev, args.steps))
                bad_rows = [k for k, v in out["extra"].items() if isinstance(v, dict) and "parity" in v and not v["parity"].get("ok")]
                if bad_rows:
                    out["extra_rows_out_of_tolerance"] = bad_rows
            except Exception as e:      # the contract line must survive a failing secondary row
                out["extra"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(512, args.bounded, timed=(p, r))
            pv = out["cpu_baseline"].get("parity_vs_gpu")
            if pv is not None and pv.get("asserted") and not pv["ok"]:
                out["config"]["finite"] = ok = False
        if saved_stdout is not None:
            sys.stdout.flush()
            # (the banner sits in the C library's stdout buffer, not yet on the descriptor: flushed now it goes where
            # descriptor 1 still points -- stderr --; flushed at exit it would follow the JSON line onto stdout)
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except OSError:
                pass
            os.dup2(saved_stdout, 1)
        # the FULL record (every `extra` row with its parity dict, the reference probe, the prose) goes to stderr and to a side
        # file; stdout carries ONE line of at most CONTRACT_LINE_MAX bytes (the 31.8 KB line of round 5 did not parse)
        full = json.dumps(out)
        sys.stderr.write("bench.py full record: " + full + "\n")
        sys.stderr.flush()
        for d in (os.environ.get("MPC_BENCH_RECORD_DIR"), os.path.join(ROOT, "gpurun_out"), ROOT):
            if not d:
                continue
            try:
                os.makedirs(d, exist_ok=True)
                with open(os.path.join(d, "bench_full.json"), "w") as fh:
                    fh.write(full + "\n")
                break
            except OSError:
                continue
        print(contract_line(out), flush=True)
        if "parity" in out and not out["parity"]["ok"]:
            sys.stderr.write("bench.py: the timed results are OUT OF TOLERANCE against the oracle: %s\n" % json.dumps(out["parity"]))
            exit_code = 4
        if out.get("extra_rows_out_of_tolerance"):
            sys.stderr.write("bench.py: secondary rows OUT OF TOLERANCE against the oracle: %s\n" % out["extra_rows_out_of_tolerance"])
            exit_code = 4
        pv = out.get("cpu_baseline", {}).get("parity_vs_gpu")
        if pv is not None and pv.get("asserted") and not pv["ok"]:
            sys.stderr.write("bench.py: the timed results are OUT OF TOLERANCE against the unmodified reference: %s\n" % json.dumps(pv))
            exit_code = 4
        if not ok and dist is not None:
            exit_code = 4
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()
