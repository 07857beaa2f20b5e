#!/bin/bash
# quick A/B on the GPU box: headline step, unbounded and box-constrained, + the parity tests that touch the dpp16 kernel
# usage: tools/quick_ab.sh TAG [lib.so ...]   (extra libraries are timed through MPC_LQR_HIP_LIB)
TAG=${1:-ab}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for lib in default "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" != default ]; then export MPC_LQR_HIP_LIB=$PWD/$lib; else unset MPC_LQR_HIP_LIB; fi
  for rep in 1 2; do
    timeout 120 python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name unbounded kernel_ms %.5f ms_per_step %.5f frac %.4f' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac']))" | tee -a $OUT/ab.log
    timeout 120 python bench.py --bounded --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name bounded   kernel_ms %.5f ms_per_step %.5f frac %.4f' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac']))" | tee -a $OUT/ab.log
  done
done
unset MPC_LQR_HIP_LIB
if [ -z "$NO_TESTS" ]; then
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "lqr_step_parity or headline or north_star or kkt or graph or smoke or masked or config5" 2>&1 | tail -5 | tee -a $OUT/ab.log
fi
