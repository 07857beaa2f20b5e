# A/B of library variants on one box: bash tools/ab_variants.sh "<bench args>" lib1.so lib2.so ...
ARGS=$1; shift
for rep in 1 2; do
for L in "$@"; do
  MPC_LQR_HIP_LIB=$PWD/$L python bench.py --no-cpu-baseline $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', 'kernel_us', round(d['roofline']['kernel_ms']*1e3,1), 'us_per_step', round(d['ms_per_step']*1e3,1))"
done; done
