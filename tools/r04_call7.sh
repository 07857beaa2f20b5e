cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_call7
for rep in 1 2; do for L in default variants/kf_nt.so; do
  if [ "$L" != default ]; then export MPC_LQR_HIP_LIB=$PWD/$L; else unset MPC_LQR_HIP_LIB; fi
  echo "$L $(python tools/prof_one.py kkt 60 150 2>/dev/null | tail -1) | $(python tools/prof_one.py kkt_bounded 60 150 2>/dev/null | tail -1)" | tee -a gpurun_out/r04_call7/ab_kkt_nt.log
done; done
