#!/usr/bin/env python3
"""Child process of bench.py's `cpu_baseline`: times the UNMODIFIED reference (locuslab/mpc.pytorch) on the host cores.

    python tools/ref_cpu_child.py REF_DIR IN.pt OUT.pt

It runs in its own interpreter because the reference's package is called `mpc`, like this repository's mirror of it: here
only REF_DIR is on the path, nothing of this repository is imported (and nothing is written next to the reference:
PYTHONDONTWRITEBYTECODE).  IN.pt: dict(x_init, C, c, F, f, cur_x, cur_u, u_lower, u_upper, reps) of CPU tensors / floats.
The timed call is one `LQRStep(...)(x_init, C, c, F, f)` = LQRStepFn.forward (mpc/lqr_step.py:277-309), all host threads
(SURVEY.md 8d), one warm call + `reps` timed ones; OUT.pt gets the median, the per-call times and the call's results."""
import os
import sys
import time

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import torch


def main():
    ref, src, dst = sys.argv[1:4]
    sys.path.insert(0, ref)
    from mpc import mpc as rmpc
    from mpc.lqr_step import LQRStep
    assert os.path.abspath(rmpc.__file__).startswith(os.path.abspath(ref)), "not the reference's mpc package: %s" % rmpc.__file__
    z = torch.load(src)
    torch.set_num_threads(os.cpu_count() or 1)
    T, B, n = z["C"].shape[0], z["C"].shape[1], z["C"].shape[2]
    ns = z["x_init"].shape[1]
    f = z["f"] if z["f"] is not None else torch.Tensor()
    cost, dx = rmpc.QuadCost(z["C"], z["c"]), rmpc.LinDx(z["F"], z["f"])
    times, out = [], None
    for rep in range(int(z.get("reps", 3)) + 1):
        step = LQRStep(n_state=ns, n_ctrl=n - ns, T=T, u_lower=z["u_lower"], u_upper=z["u_upper"], true_cost=cost,
                       true_dynamics=dx, delta_space=True, current_x=z["cur_x"], current_u=z["cur_u"])
        t0 = time.perf_counter()
        with torch.no_grad():
            out = step(z["x_init"], z["C"], z["c"], z["F"], f)
        if rep:
            times.append(time.perf_counter() - t0)
    cpu = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    times.sort()
    torch.save(dict(seconds=times[len(times) // 2], all_seconds=times, new_x=out[0], new_u=out[1], costs=out[3],
                    cores=os.cpu_count(), threads=torch.get_num_threads(), cpu_model=cpu, torch=str(torch.__version__),
                    reference=os.path.abspath(ref)), dst)


if __name__ == "__main__":
    main()
