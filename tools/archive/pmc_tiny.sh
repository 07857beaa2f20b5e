# per-wave SQ counters of the one-lane-per-problem kernels under the simulator iLQR (tools/bench_ilqr_env.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_tiny; rm -rf $O; mkdir -p $O
timeout 150 rocprofv3 --pmc SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES -d $O -o p -- python tools/bench_ilqr_env.py --kernel-only > $O/log 2>&1
python - "$O" <<'PY'
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
r = {}
for k, c, n, v in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
    if "tiny" in k: r.setdefault(k.split("(")[0][-30:], {})[c] = v
for k, v in r.items():
    w = max(v.get("SQ_WAVES", 1), 1)
    print(k, {c: round(x / w) for c, x in sorted(v.items())}, "waves", round(w))
PY
find $O -name "*.db" -delete
