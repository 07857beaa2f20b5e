cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_call11
timeout 2400 python -m pytest tests -m gpu -v 2>&1 | grep -v "PASSED" > gpurun_out/r04_call11/tests_full.log
head -c 9000 gpurun_out/r04_call11/tests_full.log | cut -c1-250
