# round 5: A/B of the 32/8 kernels over library variants (variants/<name>.so; "tree" = the working tree's library), interleaved repetitions
cd $GRAFT_REPO_ROOT
KINDS=${KINDS:-cfg5 cfg5_bare cfg5_bounded cfg5_kkt}
VARS=${VARS:-tree prev}
for rep in 1 2; do for k in $KINDS; do for v in $VARS; do L=$PWD/variants/$v.so; [ $v = tree ] && L=$PWD/mpc.pytorch_amd/mpc/libmpc_lqr_hip.so; echo -n "$v "; MPC_LQR_HIP_LIB=$L python tools/prof_one.py $k 60 200 2>&1 | tail -1 | sed 's/B=1024 ns=32 nc=8 T=64 reps=60 warm=200//'; done; done; done
