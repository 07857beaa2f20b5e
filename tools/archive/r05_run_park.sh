cd $GRAFT_REPO_ROOT
bash tools/r05_ab.sh r05_ab_park "bounded bounded_warm" default variants/prev.so 2>&1 | tail -16
echo "--- default"; python tools/r05_iter_probe.py 6 2>&1 | tail -6 | cut -c1-160
echo "--- prev"; MPC_LQR_HIP_LIB=$PWD/variants/prev.so python tools/r05_iter_probe.py 6 2>&1 | tail -6 | cut -c1-160
