# per-wave SQ counters of every lqr_step dispatch of tools/ls_probe.py (diagnostic)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/ls_pmc; rm -rf $O; mkdir -p $O
timeout 150 rocprofv3 --pmc SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O -o p -- env MPC_LQR_HIP_LIB=${LIB:-} python tools/ls_probe.py > $O/log 2>&1
python - "$O" <<'PY'
import glob, sqlite3, sys, collections
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
rows = collections.OrderedDict()
for did, k, c, v in con.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection order by dispatch_id"):
    if "lqr_step" in k: rows.setdefault(did, {})[c] = rows.setdefault(did, {}).get(c, 0) + v
seen = None
for did, r in rows.items():
    key = tuple(round(r[c] / 1024) for c in sorted(r))
    if key != seen: print(did, {c: round(r[c] / 1024) for c in sorted(r)})
    seen = key
PY
tail -30 $O/log | grep -A40 "^{"
find $O -name "*.db" -delete
