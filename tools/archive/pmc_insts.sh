# instruction counts of the LQR-step kernel for library variants: bash tools/pmc_insts.sh "<bench args>" lib1.so lib2.so ...
ARGS=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for L in "$@"; do
  O=gpurun_out/pmc_$(basename $L .so); rm -rf $O; mkdir -p $O
  MPC_LQR_HIP_LIB=$PWD/$L timeout 150 rocprofv3 --pmc ${PMC:-SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY} -d $O -o p -- python bench.py --no-cpu-baseline --steps 5 --warmup 1 $ARGS > $O/log 2>&1
  python - "$O" "$L" <<'PY'
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
r = {}
for k, c, n, v in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
    if "lqr_step" in k: r[c] = v
print(sys.argv[2], {k: round(v / 1024) for k, v in sorted(r.items())})
PY
  find $O -name "*.db" -delete
done
