cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_call1
bash tools/r04_ab_bounded.sh r04_call1 variants/base_r03.so variants/keepkm.so default 2>&1 | tail -8
MPC_LQR_HIP_LIB=$PWD/variants/prof.so timeout 300 python tools/prof_phases.py > gpurun_out/r04_call1/prof_phases.log 2>&1; tail -12 gpurun_out/r04_call1/prof_phases.log
PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" bash tools/pmc_insts.sh "--bounded --no-extra" mpc.pytorch_amd/mpc/libmpc_lqr_hip.so 2>&1 | tail -3 | tee gpurun_out/r04_call1/pmc_bounded.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "lqr_step_parity or headline or north_star or masked or ties" 2>&1 | tail -4 | tee gpurun_out/r04_call1/tests.log
