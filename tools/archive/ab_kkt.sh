for rep in 1 2; do for L in "$@"; do echo -n "$L "; MPC_LQR_HIP_LIB=$PWD/$L python tools/bench_extra.py 2>/dev/null | grep kkt_backward | tr -d '\n'; echo; done; done
