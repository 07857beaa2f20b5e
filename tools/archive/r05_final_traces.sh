#!/bin/bash
# round 5, final tree: the traces of the two headline-shape solves, the phase clocks of the 12/4 kernel, the two probes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_final_traces; mkdir -p $O
for k in bounded unbounded; do
  rm -rf /tmp/tr_$k
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$k -o tr -- python tools/trace_mpc_forward.py $k > /dev/null 2>&1
  python tools/trace_mpc_forward.py --read /tmp/tr_$k | cut -c1-150 > $O/trace_mpc_forward_$k.txt
  tail -1 $O/trace_mpc_forward_$k.txt
done
bash tools/r05_ls_cost.sh > $O/ls_cost_and_phases.log 2>&1
python tools/r05_promises_probe.py 2>&1 | grep -v amdgpu > $O/promises_probe.log
tail -3 $O/promises_probe.log
