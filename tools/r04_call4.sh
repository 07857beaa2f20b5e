cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_call4
AB_ARGS="--seed 5" bash tools/r04_ab_bounded.sh r04_call4 variants/base_r03.so default 2>&1 | tail -4
AB_ARGS="--seed 1000" bash tools/r04_ab_bounded.sh r04_call4 variants/base_r03.so default 2>&1 | tail -4
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > gpurun_out/r04_call4/tests.log 2>&1
tail -18 gpurun_out/r04_call4/tests.log
