#!/bin/bash
# per-kernel durations of the NNDynamics path under rocprofv3 --kernel-trace --stats (tools/nn_bench.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/nnprof; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o nn -- python tools/nn_bench.py "$@" > $O/log 2>&1
python - <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/nnprof/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
out = []
for r in rows[:14]:
    out.append({"name": r["Name"][:110], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])})
    print("%-110s %6d %10.1f us %8.2f ms %5.1f%%" % (out[-1]["name"], out[-1]["calls"], out[-1]["avg_us"], out[-1]["total_ms"], out[-1]["pct"]))
json.dump({"command": "rocprofv3 --kernel-trace --stats -- python tools/nn_bench.py", "kernels": out}, open("gpurun_out/nn_prof_stats.json", "w"), indent=1)
PY
tail -3 $O/log
rm -rf $O
