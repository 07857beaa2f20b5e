#!/bin/bash
# round 6, final tree (on the GPU box): rocprofv3 summaries per call kind (kernel trace with avg_us + sustained_avg_us, FETCH/WRITE,
# the SQ groups: separate --pmc passes, tools/prof_any.sh), the traces of the two headline-shape solves, the iteration probe
#     bash tools/r06_evidence.sh [KIND ...]        -> gpurun_out/r06_prof_<kind>/summary.json, gpurun_out/r06_traces/
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
KINDS=${@:-headline_alt headline bounded bounded_warm kkt cfg5 cfg5_bounded cfg5_kkt pad12_10_3 pad12_10_3_bounded}
for k in $KINDS; do
  PMC_SQ=1 bash tools/prof_any.sh r06_prof_$k python tools/prof_one.py $k 40 160 > gpurun_out/r06_prof_$k.log 2>&1
  python - $k <<'PY'
import json, sys
k = sys.argv[1]
d = json.load(open("gpurun_out/r06_prof_%s/summary.json" % k))
t = d["kernel_trace_stats"][0]
pm = d["pmc_avg_per_dispatch"]
print(k, t["name"][:50], "calls", t["calls"], "avg %.1f" % t["avg_us"], "sustained %.1f" % t.get("sustained_avg_us", -1), {kk: round(v.get("hbm_bytes_per_dispatch", 0) / 1e6, 1) for kk, v in pm.items()})
PY
done
O=gpurun_out/r06_traces; mkdir -p $O
for k in bounded unbounded; do
  rm -rf /tmp/tr_$k
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$k -o tr -- python tools/trace_mpc_forward.py $k > /dev/null 2>&1
  python tools/trace_mpc_forward.py --read /tmp/tr_$k | cut -c1-150 > $O/trace_mpc_forward_$k.txt
  tail -1 $O/trace_mpc_forward_$k.txt
done
python tools/iter_probe.py 6 2>&1 | grep iteration > $O/iter_probe_bounded.log; cat $O/iter_probe_bounded.log
