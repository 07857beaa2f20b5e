#!/usr/bin/env python3
"""Config 5 (n_state=32, n_ctrl=8, T=64), library variants interleaved on one box: the vouched step (unconstrained /
box-constrained), the sweep alone and the fused KKT backward.   python tools/ab_cfg5b.py [B] lib1.so lib2.so ...
("default" = the in-tree library; run with --child internally)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
    import torch, bench
    from mpc import _native
    from mpc._native import StepOptions
    be = _native.HipBackend()
    B = int(sys.argv[2])
    p = bench.make_problem(32, 8, 64, B, torch.float32, "cuda:0", seed=9, u_scale=0.3, clamp=1.0)
    a = (p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"])
    o = StepOptions(nominal_on_dynamics=True, c_symmetric=True)
    ob = StepOptions(u_lower=-1.0, u_upper=1.0, nominal_on_dynamics=True, c_symmetric=True)
    r = be.lqr_step(*a, o)
    gx, gu = torch.randn_like(r["new_x"]), torch.randn_like(r["new_u"])
    nx, nu = r["new_x"].clone(), r["new_u"].clone()
    out = {}
    # cold translations: a 16 us kernel that touches one byte in every 4 KiB page of 800 MB in front of every launch
    big = torch.zeros(800 * 1024 * 1024, dtype=torch.uint8, device="cuda:0")
    view = big[::4096]
    _, t0, _ = bench.timed(lambda: view.sum(), 30, 8)
    for name, fn in (("step_cold_tlb", lambda: (view.sum(), be.lqr_step(*a, o))[1]), ("step_bounded_cold_tlb", lambda: (view.sum(), be.lqr_step(*a, ob))[1])):
        _, ms, _ = bench.timed(fn, 30, 8)
        out[name] = [round((ms - t0) * 1e3, 1)]
    for rep in range(2):
        for name, fn in (("step", lambda: be.lqr_step(*a, o)), ("sweep", lambda: be.lqr_sweep(*a[:4], a[5], a[6], o)),
                         ("step_bounded", lambda: be.lqr_step(*a, ob)), ("sweep_bounded", lambda: be.lqr_sweep(*a[:4], a[5], a[6], ob)),
                         ("kkt_fused", lambda: be.kkt_backward(p["C"], p["c"], p["F"], p["f"], nx, nu, gx, gu, o)),
                         ("kkt_fused_bounded", lambda: be.kkt_backward(p["C"], p["c"], p["F"], p["f"], nx, nu, gx, gu, ob))):
            _, ms, _ = bench.timed(fn, 30, 8)
            out.setdefault(name, []).append(round(ms * 1e3, 1))
    print(json.dumps(out))
    sys.exit(0)
B = sys.argv[1]
for rep in range(2):
    for L in sys.argv[2:]:
        env = dict(os.environ)
        if L != "default":
            env["MPC_LQR_HIP_LIB"] = os.path.join(ROOT, L)
        else:
            env.pop("MPC_LQR_HIP_LIB", None)
        r = subprocess.run([sys.executable, __file__, "--child", B], env=env, capture_output=True, text=True)
        print(L, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
