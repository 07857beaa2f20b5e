#!/bin/bash
# round 5, first GPU call: the new qp_start tests, the whole -m gpu suite, the bench line, and the cold / warm A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run1; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "qp_start" 2>&1 | tail -25 ) > $O/qs_tests.log 2>&1
tail -6 $O/qs_tests.log
bash tools/r04_ab_kind.sh r05_run1 "bounded bounded_warm cfg5_bounded cfg5_bounded_warm headline" default variants/noqs.so 2>&1 | tail -32
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/tests.log 2>&1
tail -6 $O/tests.log
( time python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2>&1 | tail -3; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_run1/bench.json").read().strip().splitlines()[-1])
print("value %.4g ms_per_step %.5f frac %.4f frac_all %.4f parity %s same_set %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_all_launches"], d["parity"]["ok"], d["roofline"].get("same_set")))
for k, v in d.get("extra", {}).items():
    if isinstance(v, dict):
        print("  %-42s ms %-9s all %-9s frac %-7s parity %s %s" % (k, "%.4f" % v["ms"] if "ms" in v else "-", "%.4f" % v["ms_all_launches"] if "ms_all_launches" in v else "-", "%.3f" % v["roofline"]["frac"] if "roofline" in v else "-", v["parity"].get("ok") if "parity" in v else "-", v.get("qp_iterations_per_timestep", "")))
    else:
        print("  ", k, v)
print("bad rows:", d.get("extra_rows_out_of_tolerance"))
for k in ("cfg2_ilqr_pendulum_10iter", "cfg3_ilqr_cartpole_10iter", "nn_get_traj", "nn_linearize", "nn_rollout_linesearch", "nn_mpc_forward_5iter", "cfg5_step_B8192"):
    print(k, json.dumps(d["extra"].get(k, {}).get("parity"))[:600])
PY
tail -3 $O/bench.err
