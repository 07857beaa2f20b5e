cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_call9
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "padded" 2>&1 | tail -15 | tee gpurun_out/r04_call9/tests.log
cat gpurun_out/pad_times.json 2>/dev/null
