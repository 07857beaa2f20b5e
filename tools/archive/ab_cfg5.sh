for L in "$@"; do echo $L; MPC_LQR_HIP_LIB=$PWD/$L python tools/bench_cfg5.py ${CFG5_B:-1024} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   step', round(d['lqr_step_ms'],3), 'sweep', round(d['sweep_only_ms'],3), 'split', round(d['sweep_plus_rollout_split_ms'],3))"; done
