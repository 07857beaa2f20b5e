# do back-to-back launches of the 32/8 step kernel overlap?  kernel-trace timestamps of 24 launches
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b/overlap; rm -rf $O; mkdir -p $O
cat > /tmp/ov.py <<'PY'
import os, sys
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, ROOT)
import torch, bench
from mpc import _native
from mpc._native import StepOptions
be = _native.HipBackend()
p = bench.make_problem(32, 8, 64, 1024, torch.float32, "cuda:0", seed=5, u_scale=0.0)
o = StepOptions(nominal_on_dynamics=True, c_symmetric=True)
small = torch.zeros(1024, device="cuda:0")
for i in range(24):
    be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], o)
torch.cuda.synchronize()
for i in range(12):
    small.add_(1.0)
    be.lqr_step(p["x_init"], p["C"], p["c"], p["F"], p["f"], p["cur_x"], p["cur_u"], o)
torch.cuda.synchronize()
PY
timeout 200 rocprofv3 --kernel-trace -d $O/kt -o kt -- python /tmp/ov.py > $O/kt.log 2>&1
python - "$O" <<'PY'
import glob, sqlite3, sys
con = sqlite3.connect(glob.glob(sys.argv[1] + "/kt/**/*.db", recursive=True)[0])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table' or type='view'")]
t = [x for x in tabs if "kernel_dispatch" in x and x.startswith("rocpd_kernel_dispatch")] or [x for x in tabs if "kernel" in x]
print(t[:5])
try:
    rows = list(con.execute("select name, start, end from kernels order by start"))
except Exception as e:
    print("no kernels view:", e); rows = []
prev_end = None
for n, s, e in rows:
    if "mfma40" in n or "add" in n.lower() or "elementwise" in n:
        print(("%-28s" % n[-60:-32]), "dur_us %8.1f" % ((e - s) / 1e3), "gap_to_prev_end_us %8.1f" % (((s - prev_end) / 1e3) if prev_end else 0.0))
        prev_end = e
PY
find $O -name "*.db" -delete
