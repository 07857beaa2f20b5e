# Config 5's fused KKT backward (mpc_lqr_kkt_fused at 32/8): library variants interleaved on one box
#   bash tools/ab_kkt40_phases.sh [B] lib1.so lib2.so ...      ("default" = the in-tree library)
B=${1:-1024}; shift
for rep in 1 2; do
for L in "$@"; do
  if [ "$L" != default ]; then export MPC_LQR_HIP_LIB=$PWD/$L; else unset MPC_LQR_HIP_LIB; fi
  echo "$L $(python tools/ab_kkt.py 2 32 8 64 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v for k,v in d['us_per_backward'].items() if 'fused' in k})")"
done; done
