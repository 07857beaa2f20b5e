import sys, time, torch
sys.path[:0]=["mpc.pytorch_amd","."]
import bench
from mpc import _native
from mpc._native import StepOptions
be=_native.HipBackend()
p=bench.make_problem(12,4,50,4096,torch.float32,"cuda:0",seed=5)
opts=StepOptions()
r=be.lqr_step(p["x_init"],p["C"],p["c"],p["F"],p["f"],p["cur_x"],p["cur_u"],opts)
gx,gu=torch.randn_like(r["new_x"]),torch.randn_like(r["new_u"])
nx,nu=r["new_x"].clone(),r["new_u"].clone()
for _ in range(20): g=be.kkt_backward(p["C"],p["c"],p["F"],p["f"],nx,nu,gx,gu,opts)
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(50): g=be.kkt_backward(p["C"],p["c"],p["F"],p["f"],nx,nu,gx,gu,opts)
t1=time.perf_counter()
torch.cuda.synchronize()
t2=time.perf_counter()
print("kkt_backward: host %.1f us per call (enqueue only), total %.1f us per call" % ((t1-t0)/50*1e6,(t2-t0)/50*1e6))
