#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run6; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_nn.py -m gpu -q 2>&1 | tail -25 ) > $O/tests.log 2>&1
tail -8 $O/tests.log
python tools/nn_stress.py 60 2>&1 | tail -2
python tools/r05_nn_iter_probe.py 7 2>&1 | tail -8 | tee $O/nn_iter_probe.log
for k in nn; do
  rm -rf $O/tr_$k
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$k -o tr -- python tools/trace_mpc_forward.py $k > $O/tr_$k.log 2>&1
  python tools/trace_mpc_forward.py --read $O/tr_$k > $O/trace_$k.txt 2>&1
  tail -32 $O/trace_$k.txt | cut -c1-100
  rm -rf $O/tr_$k
done
