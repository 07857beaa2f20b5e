cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_stress
timeout 600 python tools/stress_parity.py 2048 12,4,50 3 2>/dev/null | tee gpurun_out/r04_stress/stress_12_4.log | grep -c VIOLATION
timeout 600 python tools/stress_parity.py 512 13,4,30 7 2>/dev/null | tee gpurun_out/r04_stress/stress_13_4.log | grep -c VIOLATION
timeout 600 python tools/stress_parity.py 512 24,8,30 7 2>/dev/null | tee gpurun_out/r04_stress/stress_24_8.log | grep -c VIOLATION
timeout 600 python tools/stress_parity.py 512 32,8,40 5,7 2>/dev/null | tee gpurun_out/r04_stress/stress_32_8.log | grep -c VIOLATION
for f in gpurun_out/r04_stress/*.log; do echo $f; python - $f <<'PY'
import json, sys
rows=[json.loads(l.split("   <--")[0]) for l in open(sys.argv[1]) if l.startswith("{")]
print("  rows", len(rows), "max_err_x %.2e max_err_u %.2e" % (max(r["max_err_x"] for r in rows), max(r["max_err_u"] for r in rows)), "over_tol", sum(r["over_tol"] for r in rows), "ties", sum(r["alpha_flips_or_ties"] for r in rows), "unconverged", sum(r["unconverged"] for r in rows), "nonfinite", sum(r["nonfinite"] for r in rows))
PY
done
