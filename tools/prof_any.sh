# rocprofv3 kernel trace + HBM-traffic counters (separate --pmc passes: FETCH_SIZE, WRITE_SIZE) of any command, summarised
# per kernel into one JSON:   bash tools/prof_any.sh TAG 'python tools/ab_kkt.py 1'      (on the GPU box)
TAG=$1; shift
CMD="$@"
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $CMD > $O/kt.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- $CMD > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o p -- $CMD > $O/pmc_write.log 2>&1
python - "$O" <<'PY'
import glob, json, sqlite3, sys
src = sys.argv[1]
out = {"command": open(src + "/kt.log").read()[-0:0]}
con = sqlite3.connect(glob.glob(src + "/kt/**/*.db", recursive=True)[0])
out["kernel_trace_stats"] = [dict(zip(("name", "calls", "total_us", "avg_us", "pct"), r)) for r in con.execute("select * from top_kernels limit 12")]
for r in out["kernel_trace_stats"]:
    r["name"] = r["name"][:110]
pm = {}
for db in sorted(glob.glob(src + "/pmc*/**/*.db", recursive=True)):
    con = sqlite3.connect(db)
    for k, c, n, v in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        if "mpclqr" in k:
            import re
            short = re.sub(r"\(anonymous namespace\)::|mpclqr::|void ", "", k)
            short = re.sub(r"\(.*$", "", short)
            pm.setdefault(short, {})[c] = v
# MI355X_MICROARCH.md, HBM traffic: FETCH_SIZE counts 32-byte units for these 16-byte-per-lane loads on gfx950 and reads
# half (x2); WRITE_SIZE in KiB-like units of 1024 B as reported.  Bytes per dispatch:
for k, v in pm.items():
    f, w = v.get("FETCH_SIZE"), v.get("WRITE_SIZE")
    if f is not None and w is not None:
        v["hbm_bytes_per_dispatch"] = (2.0 * f + w) * 1024.0
out["pmc_avg_per_dispatch"] = pm
json.dump(out, open(src + "/summary.json", "w"), indent=1)
print(json.dumps(out["kernel_trace_stats"][:6], indent=0)[:1500])
print(json.dumps(pm, indent=0)[:2500])
PY
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
