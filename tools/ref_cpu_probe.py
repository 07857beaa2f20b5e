#!/usr/bin/env python3
"""Times the UNMODIFIED reference (locuslab/mpc.pytorch, /root/reference) on the host cores of the BUILD
container -- the GPU box has no copy of it -- and writes profiles/ref_cpu_probe.json, which bench.py attaches to
its line as `cpu_baseline.reference_probe` ("other box": not the machine the GPU number comes from).

One LQRStep forward (mpc/lqr_step.py:277-309 of the reference: c_back + lqr_backward + lqr_forward) at the headline
shape n_state=12 n_ctrl=4 T=50 fp32, B = 512 and 4096, all host threads, on the same synthetic problem recipe as
bench.py.  The reference has O(B^2) terms (diag(alphas).mm(k), lqr_step.py:192), so B = 4096 is
also timed as eight chunks of 512 -- its best case on a CPU.

    PYTHONDONTWRITEBYTECODE=1 python tools/ref_cpu_probe.py
"""
import json
import os
import platform
import sys
import time

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MPC_REFERENCE", "/root/reference")
NS, NC, T = 12, 4, 50


def problem(B, seed=0, bounded=False):
    g = torch.Generator().manual_seed(seed)
    n = NS + NC
    A = torch.randn(T, B, n, n, generator=g)
    C = A.transpose(2, 3).matmul(A)
    c = torch.randn(T, B, n, generator=g)
    R = torch.eye(NS) + 0.2 * torch.randn(T - 1, B, NS, NS, generator=g) / NS ** 0.5
    S = torch.randn(T - 1, B, NS, NC, generator=g) / NS ** 0.5
    F = torch.cat((R, S), 3)
    f = 0.1 * torch.randn(T - 1, B, NS, generator=g)
    x0 = torch.randn(B, NS, generator=g)
    u = torch.zeros(T, B, NC)
    return C, c, F, f, x0, u


def main():
    sys.path.insert(0, REF)
    from mpc import mpc as rmpc, util as rutil
    from mpc.lqr_step import LQRStep
    torch.set_num_threads(os.cpu_count())
    rows = []
    for bounded in (False, True):
        for B, chunk in ((512, 512), (4096, 4096), (4096, 512)):
            C, c, F, f, x0, u = problem(B)
            ts = []
            for rep in range(2):
                t0 = time.perf_counter()
                for lo in range(0, B, chunk):
                    sl = slice(lo, lo + chunk)
                    dx = rmpc.LinDx(F[:, sl], f[:, sl])
                    cost = rmpc.QuadCost(C[:, sl], c[:, sl])
                    x = rutil.get_traj(T, u[:, sl], x_init=x0[sl], dynamics=dx)
                    step = LQRStep(n_state=NS, n_ctrl=NC, T=T, u_lower=-1.0 if bounded else None,
                                   u_upper=1.0 if bounded else None, true_cost=cost, true_dynamics=dx,
                                   delta_space=True, current_x=x, current_u=u[:, sl])
                    with torch.no_grad():
                        step(x0[sl], C[:, sl], c[:, sl], F[:, sl], f[:, sl])
                ts.append(time.perf_counter() - t0)
            best = min(ts)
            rows.append({"B": B, "chunk": chunk, "bounded": bounded, "seconds_per_step": round(best, 4),
                         "problem_steps_per_s": round(B * T / best, 1)})
            print(rows[-1], flush=True)
    cpu = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    out = {"what": "UNMODIFIED reference LQRStep forward (get_traj + LQRStepFn.forward), CPU tensors, fp32, "
                   "n_state=12 n_ctrl=4 T=50, timed in the BUILD container (another box than the GPU box)",
           "box": "other", "cpu_model": cpu, "cores": os.cpu_count(), "torch_threads": torch.get_num_threads(),
           "torch": torch.__version__, "python": platform.python_version(), "rows": rows,
           "script": "tools/ref_cpu_probe.py"}
    json.dump(out, open(os.path.join(ROOT, "profiles", "ref_cpu_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
