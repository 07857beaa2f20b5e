cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_call2
python tools/ab_cfg5b.py 1024 variants/base_r03.so default 2>&1 | tee gpurun_out/r04_call2/ab_cfg5.log
bash tools/r04_ab_bounded.sh r04_call2 variants/copy16.so default 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "config5 or cfg5" 2>&1 | tail -4 | tee gpurun_out/r04_call2/tests.log
