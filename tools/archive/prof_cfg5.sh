# rocprofv3 kernel trace + two PMC passes of the config-5 step (gpurun -- 'bash tools/prof_cfg5.sh')
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_cfg5
mkdir -p $O
timeout 150 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python tools/bench_cfg5.py 1024 > $O/kt.log 2>&1
timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM -d $O/pmc1 -o pmc1 -- python tools/bench_cfg5.py 1024 > $O/pmc1.log 2>&1
timeout 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM -d $O/pmc2 -o pmc2 -- python tools/bench_cfg5.py 1024 > $O/pmc2.log 2>&1
python - <<PY
import glob, json, sqlite3
O="$O"
out={}
con=sqlite3.connect(glob.glob(O+"/kt/**/*.db",recursive=True)[0])
out["kernel_trace_stats"]=[dict(zip(("name","calls","total_us","avg_us","pct"),r)) for r in con.execute("select * from top_kernels limit 6")]
for db in sorted(glob.glob(O+"/pmc*/**/*.db",recursive=True)):
    con=sqlite3.connect(db)
    for r in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        if "mfma40" in r[0]:
            out.setdefault("pmc_avg_per_dispatch",{}).setdefault(r[0].split("(")[0][-40:],{})[r[1]]=r[3]
json.dump(out,open(O+"/summary.json","w"),indent=1)
print(json.dumps(out,indent=1))
PY
find $O -name "*.db" -delete
