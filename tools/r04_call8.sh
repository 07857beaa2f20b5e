cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_call8
for rep in 1 2; do for L in default variants/base_r03.so; do
  if [ "$L" != default ]; then export MPC_LQR_HIP_LIB=$PWD/$L; else unset MPC_LQR_HIP_LIB; fi
  for k in kkt kkt_bounded cfg5_kkt cfg5_kkt_bounded; do echo "$L $(python tools/prof_one.py $k 40 120 2>/dev/null | tail -1)" | tee -a gpurun_out/r04_call8/ab_kkt_nt_all.log; done
done; done
unset MPC_LQR_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "kkt or grad or backward" 2>&1 | tail -3 | tee gpurun_out/r04_call8/tests.log
