#!/bin/bash
# round 5: what the trials after the full step cost the box-constrained 12/4 step (max_linesearch_iter 1 / 2 / 10), then the phase clocks
cd $GRAFT_REPO_ROOT
for k in bounded bounded_warm cfg5_bounded; do for m in 1 2 10; do echo -n "maxls $m: "; PROF_ONE_MAXLS=$m python tools/prof_one.py $k 60 200 2>&1 | tail -1; done; done
MPC_LQR_HIP_LIB=$PWD/variants/prof.so python tools/prof_phases.py 2>&1 | grep -v amdgpu.ids | cut -c1-700
