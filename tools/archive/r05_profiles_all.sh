#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/r05_profiles.sh ${KINDS:-headline headline_alt bounded bounded_warm cfg5 cfg5_bounded cfg5_bounded_warm kkt cfg5_kkt}
for k in ${KINDS:-headline headline_alt bounded bounded_warm cfg5 cfg5_bounded cfg5_bounded_warm kkt cfg5_kkt}; do
  python - $k <<'PY'
import json, sys
k = sys.argv[1]
d = json.load(open("gpurun_out/r05_prof_%s/summary.json" % k))
t = d["kernel_trace_stats"][0]
pm = d["pmc_avg_per_dispatch"]
print(k, t["name"][:50], "calls", t["calls"], "avg %.1f" % t["avg_us"], "sustained %.1f" % t.get("sustained_avg_us", -1), {kk: round(v.get("hbm_bytes_per_dispatch", 0) / 1e6, 1) for kk, v in pm.items()})
PY
done
