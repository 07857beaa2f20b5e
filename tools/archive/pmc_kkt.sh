# per-wave SQ counters + traffic of the KKT backward kernels (tools/bench_extra.py): bash tools/pmc_kkt.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# (one counter set: a second pass with FETCH_SIZE / WRITE_SIZE never came back from this script on the pool)
for PMC in "SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM"; do
  O=gpurun_out/pmc_kkt; rm -rf $O; mkdir -p $O
  timeout 150 rocprofv3 --pmc $PMC -d $O -o p -- python tools/bench_extra.py > $O/log 2>&1
  python - "$O" <<'PY'
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
r = {}
for k, c, n, v in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
    if "kkt" in k or "dpp16_kernel<1>" in k: r.setdefault(k.split("(")[0][-34:], {})[c] = round(v)
for k, v in r.items(): print(k, v)
PY
  find $O -name "*.db" -delete
done
