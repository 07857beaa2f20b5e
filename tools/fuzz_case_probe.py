"""One case of tools/emu_fuzz_kkt.py's padded-shape generator as a plain LQR step through the generic (impl 1), one-problem-per-wavefront
(impl 2) and padded 12/4 (impl 8) float32 kernels against the float64 oracle: is a fuzz violation the kernel or float32?
    python tools/fuzz_case_probe.py CASE SEED0          (on the GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd"))
from oracle import lqr_oracle as O
import torch
from mpc import _native
from mpc._native import StepOptions
be = _native.HipBackend()
dev = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(torch.float32).to("cuda:0")
case, seed0 = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed0 * 7919 + case)
ns, nc = int(rng.integers(1, 13)), int(rng.integers(1, 5))
n = ns + nc
T = int(rng.choice([1, 2, 3, 4, 6, 7, 9, 17, 33, 63, 64, 65, 70]))
B = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 9, 33, 130]))
Tm = max(T, 2)
A = rng.standard_normal((Tm, B, n, n)); C = np.einsum("tbji,tbjk->tbik", A, A)
c = rng.standard_normal((Tm, B, n))
F = np.concatenate((np.eye(ns) + 0.2 * rng.standard_normal((Tm - 1, B, ns, ns)) / np.sqrt(ns), rng.standard_normal((Tm - 1, B, ns, nc)) / np.sqrt(ns)), 3)
f = 0.1 * rng.standard_normal((Tm - 1, B, ns)) if rng.random() < 0.7 else None
x_init = rng.standard_normal((B, ns))
print("shape", ns, nc, T, B, "f", f is not None)
C32, c32, F32 = C.astype(np.float32).astype(np.float64), c.astype(np.float32).astype(np.float64), F.astype(np.float32).astype(np.float64)
f32_ = None if f is None else f.astype(np.float32).astype(np.float64)
x0 = x_init.astype(np.float32).astype(np.float64)
cur_u = np.zeros((T, B, nc)); cur_x, _ = O.traj_cost(x0, cur_u, F32, f32_)
cur_x = cur_x.astype(np.float32).astype(np.float64)
o = O.lqr_step(x0, C32, c32, F32, f32_, cur_x, cur_u, lockstep=False)
for impl in (1, 2, 8):
    r = be.lqr_step(dev(x0), dev(C32), dev(c32), dev(F32), dev(f32_), dev(cur_x), dev(cur_u), StepOptions(), impl=impl)
    du = (r["new_u"].cpu().numpy() - o["new_u"])
    per = np.abs(du).max(axis=(0, 2)) / max(1.0, np.abs(o["new_u"]).max())
    print("impl", impl, "max rel err new_u %.3g at problem %d; alphas differ: %d" % (per.max(), per.argmax(), int((r["alphas"].cpu().numpy() != o["alphas"]).sum())),
          "top5", np.sort(per)[-5:])
