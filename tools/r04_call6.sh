cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_call6
for k in kkt kkt_bounded bounded headline cfg5 cfg5_bounded cfg5_kkt; do PROF_ONE_TRACE=20 python tools/prof_one.py $k 2>/dev/null | tee -a gpurun_out/r04_call6/settle_trace.log; done
