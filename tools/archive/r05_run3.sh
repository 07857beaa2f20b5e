#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run3; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_nn.py -m gpu -q -k "qp_start or nn" 2>&1 | tail -25 ) > $O/tests.log 2>&1
tail -8 $O/tests.log
bash tools/r04_ab_kind.sh r05_run3 "cfg5_bounded cfg5_bounded_warm" default variants/noqs.so 2>&1 | tail -12
for k in bounded nn; do
  rm -rf $O/tr_$k
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$k -o tr -- python tools/trace_mpc_forward.py $k > $O/tr_$k.log 2>&1
  python tools/trace_mpc_forward.py --read $O/tr_$k > $O/trace_$k.txt 2>&1
  tail -40 $O/trace_$k.txt
  rm -rf $O/tr_$k
done
python - <<'PY'
import sys, os, torch, time
sys.path.insert(0, "mpc.pytorch_amd"); sys.path.insert(0, ".")
import bench
from mpc import mpc
from mpc.mpc import QuadCost
from mpc.dynamics import NNDynamics
torch.manual_seed(0)
dyn = NNDynamics(12, 4, [100], activation="sigmoid").to("cuda:0")
p = bench.make_problem(12, 4, 50, 4096, torch.float32, "cuda:0", seed=5, u_scale=0.3, clamp=1.0)
ctrl = mpc.MPC(12, 4, 50, u_lower=-1.0, u_upper=1.0, lqr_iter=5, verbose=-1, exit_unconverged=False, detach_unconverged=False,
               grad_method=mpc.GradMethods.ANALYTIC, backprop=False, u_init=p["cur_u"].clone())
cost = QuadCost(p["C"], p["c"])
with torch.no_grad():
    for rep in range(3):
        ms, mean, mx, _ = bench.timed_each(lambda: ctrl(p["x_init"], cost, dyn), 9, 3)
        print("nn_mpc_forward_5iter: median %.3f mean %.3f max %.3f ms" % (ms, mean, mx))
PY
