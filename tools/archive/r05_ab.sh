#!/bin/bash
# round 5: A/B of library variants on prof_one.py kinds + the parity tests of the 12/4 kernel
cd $GRAFT_REPO_ROOT
TAG=$1; KINDS=$2; shift 2
bash tools/r04_ab_kind.sh $TAG "$KINDS" "$@" 2>&1 | tail -40
( timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q -x -k "headline or qp_start or dpp16 or step_parity or long_horizon or both_rings" 2>&1 | tail -5 )
