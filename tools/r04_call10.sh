cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_call10
for rep in 1 2; do for L in default variants/c_plain.so; do
  if [ "$L" != default ]; then export MPC_LQR_HIP_LIB=$PWD/$L; else unset MPC_LQR_HIP_LIB; fi
  for k in cfg5 cfg5_bounded cfg5_kkt cfg5_B8192; do echo "$L $(python tools/prof_one.py $k 40 120 2>/dev/null | tail -1)" | tee -a gpurun_out/r04_call10/ab_cfg5_c_nt.log; done
done; done
