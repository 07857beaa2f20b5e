cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_call3
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r04_call3/tests.log 2>&1
tail -18 gpurun_out/r04_call3/tests.log
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r04_call3/bench.json 2> gpurun_out/r04_call3/bench.err ) 2>&1 | tail -3; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_call3/bench.json").read().strip().splitlines()[-1])
print("value %.4g ms_per_step %.5f frac %.4f frac_all %.4f parity %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_all_launches"], d["parity"]["ok"]))
for k, v in d.get("extra", {}).items():
    if isinstance(v, dict):
        print("  %-42s ms %-9s frac %-7s parity %s" % (k, "%.4f" % v["ms"] if "ms" in v else "-", "%.3f" % v["roofline"]["frac"] if "roofline" in v else "-", (v["parity"].get("ok"), {a: b for a, b in v["parity"].items() if a.startswith("max_err") or a.endswith("ties") or a == "worst_rel" or a == "error"}) if "parity" in v else "-"))
print("cpu_baseline:", {k: v for k, v in d.get("cpu_baseline", {}).items() if k not in ("reference_probe", "sample")})
PY
tail -3 gpurun_out/r04_call3/bench.err
MPC_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_call3/bench_force_dist.json 2> gpurun_out/r04_call3/bench_force_dist.err; echo "force_dist rc=$?"; tail -c 1500 gpurun_out/r04_call3/bench_force_dist.json; tail -3 gpurun_out/r04_call3/bench_force_dist.err
