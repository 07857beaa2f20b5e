import os, sys, torch, numpy as np
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import bench
from mpc import _native
from mpc._native import StepOptions
be = _native.HipBackend()
p = bench.make_problem(12, 4, 50, 4096, torch.float32, "cuda:0", seed=0, u_scale=0.3, clamp=1.0)
opts = StepOptions(u_lower=-1.0, u_upper=1.0)
for impl in (1, 2, 3):
    r = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], opts, impl=impl)
    torch.cuda.synchronize()
    st = r["status"].cpu().numpy(); qp = r["qp_iters"].cpu().numpy()
    print("impl", impl, "unconverged problems %.4f" % (st & 1).mean(), "qp_iters mean %.1f max %d" % (qp.mean(), qp.max()),
          "hist>300:", int((qp > 300).sum()))
from oracle import lqr_oracle as O
h = {k: (None if v is None else v.cpu().numpy()) for k, v in p.items()}
B = 256
sl = lambda a, d: np.ascontiguousarray(a[:, :B] if d else a[:B])
o = [O.lqr_step(sl(h["x_init"],0)[b:b+1], sl(h["C"],1)[:, b:b+1], sl(h["c"],1)[:, b:b+1], sl(h["F"],1)[:, b:b+1], sl(h["f"],1)[:, b:b+1],
                sl(h["cur_x"],1)[:, b:b+1], sl(h["cur_u"],1)[:, b:b+1], -1.0, 1.0, lockstep=False)["n_qp_iter"] for b in range(B)]
print("oracle f32 per-problem qp iters: mean %.1f max %d" % (np.mean(o), np.max(o)))
r = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], opts, impl=3)
print("dpp16 first 256: mean %.1f max %d" % (r["qp_iters"][:256].float().mean().item(), r["qp_iters"][:256].max().item()))
r = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], opts, impl=1)
print("generic first 256: mean %.1f max %d" % (r["qp_iters"][:256].float().mean().item(), r["qp_iters"][:256].max().item()))
