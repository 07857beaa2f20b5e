#!/bin/bash
# round 5: the whole -m gpu suite + the bench line as the driver runs it
cd $GRAFT_REPO_ROOT
TAG=${1:-r05_full}
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/tests.log 2>&1
tail -6 $O/tests.log
( time python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2>&1 | tail -3; echo "bench rc=$?"
python - $TAG <<'PY'
import json, sys
d = json.loads(open("gpurun_out/%s/bench.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("value %.4g ms_per_step %.5f frac %.4f frac_all %.4f parity %s same_set %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_all_launches"], d["parity"]["ok"], {k: v for k, v in (d["roofline"].get("same_set") or {}).items() if k != "note"}))
for k, v in d.get("extra", {}).items():
    if isinstance(v, dict):
        print("  %-42s ms %-9s all %-9s frac %-7s parity %s %s" % (k, "%.4f" % v["ms"] if "ms" in v else "-", "%.4f" % v["ms_all_launches"] if "ms_all_launches" in v else "-", "%.3f" % v["roofline"]["frac"] if "roofline" in v else "-", v["parity"].get("ok") if "parity" in v else "-", v.get("qp_iterations_per_timestep", "")))
    else:
        print("  ", k, v)
print("bad rows:", d.get("extra_rows_out_of_tolerance"))
print("cpu_baseline:", {k: v for k, v in d.get("cpu_baseline", {}).items() if k not in ("reference_probe", "sample")})
PY
tail -3 $O/bench.err
