#!/bin/bash
# round 4: A/B of library variants on the box-constrained (and unconstrained) headline step, interleaved on one box.
#   bash tools/r04_ab_bounded.sh TAG lib1.so lib2.so ...     ("default" = the in-tree library)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do
for lib in "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" != default ]; then export MPC_LQR_HIP_LIB=$PWD/$lib; else unset MPC_LQR_HIP_LIB; fi
  for mode in "--bounded" ${AB_UNBOUNDED:+""}; do
    timeout 200 python bench.py $mode $AB_ARGS --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name ${mode:-unbounded} kernel_ms %.5f ms_per_step %.5f frac %.4f' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac']))" | tee -a $OUT/ab.log
  done
done; done
unset MPC_LQR_HIP_LIB
