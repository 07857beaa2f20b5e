#!/bin/bash
for rep in 1 2; do
for lib in default variants/nstage2.so; do
  if [ "$lib" != default ]; then export MPC_LQR_HIP_LIB=$PWD/$lib; else unset MPC_LQR_HIP_LIB; fi
  for B in 1024 2048 4096 6144; do
    for mode in "" "--bounded"; do
      timeout 200 python bench.py --no-extra --no-cpu-baseline --batch $B $mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib B=$B $mode kernel_ms %.5f frac %.4f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"
    done
  done
done
done
