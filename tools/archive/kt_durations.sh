#!/bin/bash
# per-launch durations of the LQR-step kernel under rocprofv3 --kernel-trace (how the launches settle)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/ktd; rm -rf $O; mkdir -p $O
timeout 150 rocprofv3 --kernel-trace --output-format csv -d $O -o kt -- python bench.py --no-cpu-baseline --no-extra "$@" > $O/log 2>&1
python - "$@" <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ktd/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "lqr_step" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
gap = [(int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3 for i in range(len(rows) - 1)]
import json, sys
steps = 20
for i, a in enumerate(sys.argv):
    if a == "--steps": steps = int(sys.argv[i + 1])
summ = {"command": "rocprofv3 --kernel-trace -- python bench.py --no-cpu-baseline --no-extra " + " ".join(sys.argv[1:]),
        "kernel": rows[0]["Kernel_Name"], "calls": len(d), "avg_us_all_calls": sum(d) / len(d),
        "avg_us_timed_region": sum(d[-steps:]) / steps, "timed_region": "the last %d calls (the K timed steps of bench.py)" % steps,
        "min_us": min(d), "max_us": max(d), "per_call_us": [round(x, 1) for x in d],
        "note": "launches 1-17 run at boost clocks, then the power controller dips (~100 us per launch) and settles by launch ~120"}
json.dump(summ, open("gpurun_out/r02_kt_durations.json", "w"), indent=1)
print("n", len(d), "avg all %.2f  timed region %.2f" % (summ["avg_us_all_calls"], summ["avg_us_timed_region"]))
for i in range(0, len(d), 10):
    print(i, " ".join("%.1f" % x for x in d[i:i + 10]), "| gaps", " ".join("%.1f" % x for x in gap[i:i + 10]))
PY
rm -rf $O
