#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run5; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_nn.py -m gpu -q -k "qp_start or nn" 2>&1 | tail -25 ) > $O/tests.log 2>&1
tail -8 $O/tests.log
for k in nn; do
  rm -rf $O/tr_$k
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$k -o tr -- python tools/trace_mpc_forward.py $k > $O/tr_$k.log 2>&1
  python tools/trace_mpc_forward.py --read $O/tr_$k > $O/trace_$k.txt 2>&1
  tail -32 $O/trace_$k.txt
  rm -rf $O/tr_$k
done
python tools/nn_stress.py 40 2>&1 | tail -5
