# round 5: A/B over library variants (variants/<name>.so) for any prof_one kinds, interleaved repetitions
cd $GRAFT_REPO_ROOT
KINDS=${KINDS:-headline bounded cfg5 cfg5_bounded kkt cfg5_kkt}
VARS=${VARS:-tree0 noslp}
for rep in 1 2; do for k in $KINDS; do for v in $VARS; do echo -n "$v "; MPC_LQR_HIP_LIB=$PWD/variants/$v.so python tools/prof_one.py $k 60 200 2>&1 | tail -1 | sed 's/ reps=60 warm=200//'; done; done; done
