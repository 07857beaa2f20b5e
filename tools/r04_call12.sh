cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_call12
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "kkt or fused or backward" 2>&1 | tail -6 | tee gpurun_out/r04_call12/tests.log
python - <<'PY'
# the backward at T = 100 (beyond the register-resident gains): one launch against three
import sys, os, torch
sys.path.insert(0, "mpc.pytorch_amd"); sys.path.insert(0, ".")
import bench
from mpc import _native
from mpc._native import StepOptions
be = _native.HipBackend()
for T in (64, 100):
    p = bench.make_problem(12, 4, T, 4096, torch.float32, "cuda:0", seed=5)
    r = be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], StepOptions())
    gx, gu = torch.randn_like(r["new_x"]), torch.randn_like(r["new_u"])
    nx, nu = r["new_x"].clone(), r["new_u"].clone()
    for name, o in (("fused", StepOptions(c_symmetric=True)), ("three_launch", StepOptions())):
        fn = lambda: be.kkt_backward(p["C"], p["c"], p["F"], p["f"], nx, nu, gx, gu, o)
        for _ in range(100): fn()
        _, ms, _ = bench.timed(fn, 30, 0)
        ab = bench.kkt_algorithmic_bytes_per_problem(12, 4, T) * 4096
        print("T=%d %s: %.4f ms  frac %.3f" % (T, name, ms, ab / (ms * 1e-3) / 8e12))
PY
