cd $GRAFT_REPO_ROOT
TAG=${1:-r04_bench}
mkdir -p gpurun_out/$TAG
( time python bench.py --steps 20 --warmup 5 > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err ) 2>&1 | tail -3; echo "bench rc=$?"
python - $TAG <<'PY'
import json, sys
d = json.loads(open("gpurun_out/%s/bench.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("value %.4g ms_per_step %.5f frac %.4f frac_all %.4f parity %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_all_launches"], d["parity"]["ok"]))
for k, v in d.get("extra", {}).items():
    if isinstance(v, dict):
        print("  %-42s ms %-9s all %-9s frac %-7s parity %s %s" % (k, "%.4f" % v["ms"] if "ms" in v else "-", "%.4f" % v["ms_all_launches"] if "ms_all_launches" in v else "-", "%.3f" % v["roofline"]["frac"] if "roofline" in v else "-", v["parity"].get("ok") if "parity" in v else "-", ("x%.1f over generic" % v["speedup_over_generic"]) if "speedup_over_generic" in v else ""))
PY
