cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_call13
for rep in 1 2; do for L in variants/before_mcol.so variants/dppbcast.so default; do
  if [ "$L" != default ]; then export MPC_LQR_HIP_LIB=$PWD/$L; else unset MPC_LQR_HIP_LIB; fi
  for k in bounded headline kkt; do echo "$L $(python tools/prof_one.py $k 40 150 2>/dev/null | tail -1)" | tee -a gpurun_out/r04_call13/ab.log; done
done; done
unset MPC_LQR_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "lqr_step_parity or headline or north_star or masked or ties or kkt" 2>&1 | tail -3
