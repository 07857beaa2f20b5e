#!/bin/bash
# the 12/4 kernel with a 2-slot sweep ring (variants/nstage2.so, -DMPC_DPP16_NSTAGE=2: 18 KiB of LDS per wave, eight waves per
# CU = two per SIMD) against the product (4 slots, four waves per CU), at B = 4096 / 8192 / 16384
for lib in default variants/nstage2.so; do
  if [ "$lib" != default ]; then export MPC_LQR_HIP_LIB=$PWD/$lib; else unset MPC_LQR_HIP_LIB; fi
  for B in 4096 8192 16384; do
    for mode in "" "--bounded"; do
      timeout 200 python bench.py --no-extra --no-cpu-baseline --batch $B $mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib B=$B $mode kernel_ms %.5f frac %.4f finite %s' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['finite']))"
    done
  done
done
export MPC_LQR_HIP_LIB=$PWD/variants/nstage2.so
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "lqr_step_parity or headline or north_star or masked" 2>&1 | tail -3
