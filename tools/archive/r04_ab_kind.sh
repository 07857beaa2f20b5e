#!/bin/bash
# round 4: A/B of library variants on ONE call kind of tools/prof_one.py, interleaved on one box in the sustained state.
#   bash tools/r04_ab_kind.sh TAG "KIND1 KIND2 ..." lib1.so lib2.so ...     ("default" = the in-tree library)
TAG=$1; KINDS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2 3; do
for lib in "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" != default ]; then export MPC_LQR_HIP_LIB=$PWD/$lib; else unset MPC_LQR_HIP_LIB; fi
  for kind in $KINDS; do
    echo "$name $(timeout 200 python tools/prof_one.py $kind 60 200 2>/dev/null | tail -1)" | tee -a $OUT/ab.log
  done
done; done
unset MPC_LQR_HIP_LIB
