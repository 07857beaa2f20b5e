#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run2; mkdir -p $O
python tools/r05_tie_probe.py 12 4 70 516 43 2>&1 | tail -20 | tee $O/tie_12_4.log
( time timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "qp_start" 2>&1 | tail -25 ) > $O/qs_tests.log 2>&1
tail -12 $O/qs_tests.log
